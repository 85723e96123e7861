#!/bin/bash
# round 4, call 30: what the any-hit test costs the traversal kernels of configs[3]: the same frame with the test compiled out (FH_NO_ALPHA=1: a different image, an upper bound for
# anything that makes the test cheaper)
cd $GRAFT_REPO_ROOT
bash tools/gpu_env_ab.sh "FH_X=0 FH_NO_ALPHA=1 FH_X=0 FH_NO_ALPHA=1" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
