#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== the test that aborted, alone, uncaptured"; FH_SKY_SPLIT_MIN_LOG2=0 FH_DEBUG_BVH=1 PYTHONFAULTHANDLER=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -s -q -k "test_reference_firsthit_bug_compat_mode" > gpurun_out/r4_abort.log 2>&1; echo "rc=$?"; grep -v "^\[trace\]\|^\[bvh\]\|^\[tail\]" gpurun_out/r4_abort.log | head -60 | cut -c1-250
echo "== FH_COLLAPSE=greedy mismatch"; FH_COLLAPSE=greedy timeout -k 10 120 python3 tools/debug_collapse.py 2>&1 | tail -16
