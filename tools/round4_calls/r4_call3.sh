#!/bin/bash
# round 4, call 3: occupancy decisions of the streaming kernels (FH_DEBUG_BVH), node-visit microbenchmark, issue_peak --json, LUT pin on the GPU
cd $GRAFT_REPO_ROOT
echo "== occupancy, configs[3]"; FH_DEBUG_BVH=1 timeout -k 10 200 python3 tools/sponza_probe.py 64 2>&1 | grep "\[trace\]\|^alone\|^pipelined" | cut -c1-600
echo "== occupancy, configs[2]"; CFG=2 FH_DEBUG_BVH=1 timeout -k 10 200 python3 tools/sponza_probe.py 64 2>&1 | grep "\[trace\]\|^alone\|^pipelined" | cut -c1-600
echo "== node visit"; timeout -k 10 300 tools/micro/node_visit.bin 60 2>&1 | tee gpurun_out/r4_node_visit.txt
echo "== issue peak json"; timeout -k 10 120 tools/micro/issue_peak.bin --json gpurun_out/r4_issue_peak.json $(sha256sum fredholm_amd/csrc/fh_trace.h | cut -c1-16) 60; cat gpurun_out/r4_issue_peak.json
echo "== LUT pin on the GPU"; timeout -k 10 300 python3 -m pytest tests/test_lut_integral_pin.py -m gpu -x -q -s 2>&1 | tail -8
