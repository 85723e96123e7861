#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 tools/micro/issue_peak.bin 60 > gpurun_out/r03_issue_peak.txt 2>&1; echo "issue_peak rc $?"
timeout -k 10 900 bash tools/profile_round3.sh r03_a 2 384
