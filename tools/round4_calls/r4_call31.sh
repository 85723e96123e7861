#!/bin/bash
# round 4, call 31: what stack levels in LDS are worth on configs[3] (14 levels; 7 in LDS by default): 5 / 3 levels -- the price of LDS for anything else in the streaming kernels
cd $GRAFT_REPO_ROOT
bash tools/gpu_env_ab.sh "FH_X=0 FH_STACK_LDS=5 FH_STACK_LDS=3 FH_X=0" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
