#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-330
for tp in 65536 262144; do echo "FH_TAIL_PATHS=$tp"; FH_TAIL_PATHS=$tp timeout -k 10 300 python3 tools/latency_breakdown.py 2 2>&1 | cut -c1-330; done
bash tools/gpu_ab.sh "base" "2 4 1" "--steps 4 --warmup 1 --no-extras"
FH_TAIL_PATHS=131072 bash tools/gpu_ab.sh "base" "2 4" "--steps 4 --warmup 1 --no-extras"
