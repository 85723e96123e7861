#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/c11_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c11_pytest.txt
bash tools/gpu_ab.sh "r2 base" "3 2 4 1" "--steps 2 --warmup 1 --no-extras"
