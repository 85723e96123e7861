#!/bin/bash
# round 6, call 9: suite on the build without the ray items (sort_onepass = 2 kept), the AFTER timelines of the one-pass calls, latency of all three configurations
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_9_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r6_9_tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spp in 1 16; do
  rm -rf gpurun_out/tl_$spp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$spp -o tl -- python3 tools/call_timeline.py run 3 $spp 40 > gpurun_out/r6_9_tl_run_$spp.log 2>&1 || { tail -5 gpurun_out/r6_9_tl_run_$spp.log; exit 1; }
  grep "configs\[" gpurun_out/r6_9_tl_run_$spp.log
  f=$(find gpurun_out/tl_$spp -name "*kernel_trace.csv" | head -1)
  python3 tools/call_timeline.py reduce $f $spp 40 > gpurun_out/r6_after_timeline_config3_${spp}spp.txt && tail -4 gpurun_out/r6_after_timeline_config3_${spp}spp.txt
  rm -rf gpurun_out/tl_$spp
done
timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 > gpurun_out/r6_9_latency.log 2>&1; cat gpurun_out/r6_9_latency.log
