#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/c19_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/c19_pytest.txt
timeout -k 10 500 bash tools/profile_round3.sh r03_c 2 384 | tail -2
timeout -k 10 500 bash tools/profile_round3.sh r03_c1 1 258 | tail -2
