#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/c13_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/c13_pytest.txt
bash tools/gpu_ab.sh "base" "4" "--steps 2 --warmup 1 --no-extras"
