#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_ab.sh "base noslack base noslack" "4" "--steps 2 --warmup 1 --no-extras"
bash tools/gpu_ab.sh "base noslack" "3" "--steps 1 --warmup 1 --no-extras"
