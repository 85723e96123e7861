#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/c5_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/c5_pytest.txt
for mr in 64; do echo "FH_STREAM_MIN_RAYS=$mr"; FH_STREAM_MIN_RAYS=$mr timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-330; done
echo "per-lane tail (FH_COOP=0 disables cooperative tests everywhere)"; FH_STREAM_MIN_RAYS=64 FH_COOP=0 timeout -k 10 300 python3 tools/latency_breakdown.py 2 2>&1 | cut -c1-330
for tp in 131072 262144; do echo "FH_TAIL_PATHS=$tp"; FH_STREAM_MIN_RAYS=64 FH_TAIL_PATHS=$tp timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-330; done
bash tools/gpu_ab.sh "base" "2 4" "--steps 2 --warmup 1"
