#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-330
bash tools/gpu_ab.sh "base" "2 4" "--steps 4 --warmup 1 --no-extras"
FH_COOP=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
