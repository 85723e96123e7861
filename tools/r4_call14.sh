#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== parity suite, split forced, uncaptured"; FH_SKY_SPLIT_MIN_LOG2=0 PYTHONFAULTHANDLER=1 timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -s -q > gpurun_out/r4_forced_s.log 2>&1; echo "rc=$?"; grep -v "^\[trace\]\|^\[bvh\]\|^\[tail\]\|^\[split\]" gpurun_out/r4_forced_s.log | grep -n -i "abort\|fault\|error\|terminate\|what\|free\|corrupt\|passed\|failed" | head -20
echo "== parity suite, defaults"; timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
