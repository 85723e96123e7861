#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== parity suite, FH_SKY_SPLIT=0, verbose to file"; FH_SKY_SPLIT=0 PYTHONFAULTHANDLER=1 timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -s -v > gpurun_out/r4_default_v.log 2>&1; echo "rc=$?"; grep -v "^\[trace\]\|^\[bvh\]\|^\[tail\]\|^\[split\]" gpurun_out/r4_default_v.log | grep -n "PASSED\|FAILED\|fault\|Fatal" | tail -4
