#!/bin/bash
# round 4, call 20: after the fix of the null-stream hipMemset (pool_ensure): the poison test alone first (a fault there must not be followed by anything), then the GPU
# suite as the driver runs it, then the parity file with the sky-pixel split forced on every call and with it off
cd $GRAFT_REPO_ROOT
set -o pipefail
run() { echo "== $1"; shift; "$@" > gpurun_out/r4_c20_$N.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c20_$N.log; grep -i -m2 "Memory access fault\|Fatal Python" gpurun_out/r4_c20_$N.log; [ $rc -eq 0 ] || { echo "rc=$rc: stopping"; exit 1; }; }
N=poison run "poison test" env PYTHONFAULTHANDLER=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "garbage"
N=suite run "GPU suite" env PYTHONFAULTHANDLER=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q
N=forced run "parity file, split forced" env FH_SKY_SPLIT_MIN_LOG2=0 PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q
N=off run "parity file, split off" env FH_SKY_SPLIT=0 PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q
