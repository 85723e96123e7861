#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 > gpurun_out/latency_breakdown.txt 2>&1; cat gpurun_out/latency_breakdown.txt
for tp in 65536 16384 4096 0; do echo "FH_TAIL_PATHS=$tp"; FH_TAIL_PATHS=$tp timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-400; done
