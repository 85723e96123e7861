#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/c17_pytest.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c17_pytest.txt
bash tools/gpu_ab.sh "noroute base noroute base" "2" "--steps 4 --warmup 1 --no-extras"
bash tools/gpu_ab.sh "noroute base" "3 4" "--steps 1 --warmup 1 --no-extras"
timeout -k 10 300 python3 tools/latency_breakdown.py 2 2>&1 | cut -c1-330
