#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/c4_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/c4_pytest.txt
for mr in 0 64 128 256 512; do echo "FH_STREAM_MIN_RAYS=$mr"; FH_STREAM_MIN_RAYS=$mr timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-330; done
echo "MIN_RAYS=128 TAIL_PATHS sweep"
for tp in 16384 4096 1024; do echo "FH_TAIL_PATHS=$tp"; FH_STREAM_MIN_RAYS=128 FH_TAIL_PATHS=$tp timeout -k 10 300 python3 tools/latency_breakdown.py 2 2>&1 | cut -c1-330; done
bash tools/gpu_ab.sh "base" "2 1" "--steps 4 --warmup 1"
FH_SORT_SMALL=1 bash tools/gpu_ab.sh "base" "1" "--steps 4 --warmup 1"
