#!/bin/bash
# round 4, call 28: (a) CMJ block form in k_sky_pixels: sampler KAT + split test, configs[2]; (b) next-node prefetch variant (FH_NODE_PREFETCH=1) against the default build, configs[3] and [2]
cd $GRAFT_REPO_ROOT
echo "== tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cmj or sky_pixel_split or camera" > gpurun_out/r4_c28_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c28_tests.log; [ $rc -eq 0 ] || exit 1
echo "== configs[2]"; bash tools/gpu_ab.sh "base pf base pf" "2" "--steps 6 --warmup 2 --no-extras"
echo "== configs[3]"; bash tools/gpu_ab.sh "base pf base pf" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
