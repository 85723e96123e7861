#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/c18_pytest.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/c18_pytest.txt
SPP=1024 BENCH_SIZING=1 WORLDS=1,8 timeout -k 10 300 python3 tools/shard_time.py 2>&1 | grep world
SPP=256 BENCH_SIZING=1 WORLDS=1,8 timeout -k 10 300 python3 tools/shard_time.py 2>&1 | grep world
