#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== split test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sky_pixel_split" > gpurun_out/r4_split_test.log 2>&1; tail -5 gpurun_out/r4_split_test.log
echo "== parity suite, split forced"; FH_SKY_SPLIT_MIN_LOG2=0 PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -v > gpurun_out/r4_split_forced.log 2>&1; echo "rc=$?"; grep -n "PASSED\|FAILED" gpurun_out/r4_split_forced.log | tail -3; grep -n "Fatal\|fault\|Abort\|HSA\|hip" gpurun_out/r4_split_forced.log | head -20
