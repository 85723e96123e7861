#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_ab.sh "base oct base oct" "2" "--steps 4 --warmup 1 --no-extras"
bash tools/gpu_ab.sh "base oct" "3 4" "--steps 1 --warmup 1 --no-extras"
