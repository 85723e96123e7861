#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tp in 65536 131072 262144; do echo "FH_TAIL_PATHS=$tp"; FH_TAIL_PATHS=$tp bash tools/gpu_ab.sh "base" "4" "--steps 2 --warmup 1 --no-extras"; done
for tp in 65536 131072; do echo "FH_TAIL_PATHS=$tp"; FH_TAIL_PATHS=$tp bash tools/gpu_ab.sh "base" "3" "--steps 1 --warmup 1 --no-extras"; done
