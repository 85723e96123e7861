#!/bin/bash
# round 4, call 25: sky pixels by a second, tighter test (the padded bounds themselves, grown by the pixel's ray-bundle radius): split test + poison test, configs[2] / [4] frames,
# and what small calls gain when they split as well (FH_SKY_SPLIT_MIN_LOG2)
cd $GRAFT_REPO_ROOT
echo "== tests"; FH_DEBUG_BVH=1 PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "sky_pixel_split or garbage" > gpurun_out/r4_c25_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c25_tests.log; grep "^\[split\]" gpurun_out/r4_c25_tests.log | sort | uniq -c | head -8; [ $rc -eq 0 ] || exit 1
echo "== configs[2]"; bash tools/gpu_env_ab.sh "FH_X=0 FH_X=1" "2" "--steps 6 --warmup 2 --no-extras"
echo "== configs[4], 1024 spp"; bash tools/gpu_env_ab.sh "FH_X=0" "4" "--spp 1024 --steps 2 --warmup 1 --no-extras"
for v in FH_SKY_SPLIT_MIN_LOG2=24; do echo "-- $v"; env $v timeout -k 10 300 python3 tools/latency_breakdown.py 2 2>/dev/null | cut -c1-260; done
