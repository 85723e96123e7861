#!/bin/bash
# round 3, call 1: parity suite on the 64-byte node layout, issue-rate microbenchmark with clocks, A/B against the round-2 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/c1_pytest.txt
timeout -k 10 300 tools/micro/issue_peak.bin 60 > gpurun_out/r03_issue_peak.txt 2>&1; echo "issue_peak rc $?"; head -30 gpurun_out/r03_issue_peak.txt
bash tools/gpu_ab.sh "r2 base" "2 3 1"
