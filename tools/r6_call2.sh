#!/bin/bash
# round 6, call 2: the rest of the GPU suite (call 1 stopped at a test of its own that asserted a scene-dependent decision), and the BEFORE timelines (csv output this time)
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_2_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_2_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spp in 1 16; do
  rm -rf gpurun_out/tl_$spp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$spp -o tl -- python3 tools/call_timeline.py run 3 $spp 40 > gpurun_out/r6_2_tl_run_$spp.log 2>&1 || { tail -5 gpurun_out/r6_2_tl_run_$spp.log; exit 1; }
  grep "configs\[" gpurun_out/r6_2_tl_run_$spp.log
  f=$(find gpurun_out/tl_$spp -name "*kernel_trace.csv" | head -1)
  python3 tools/call_timeline.py reduce $f $spp 40 > gpurun_out/r6_before_timeline_config3_${spp}spp.txt && tail -4 gpurun_out/r6_before_timeline_config3_${spp}spp.txt
  rm -rf gpurun_out/tl_$spp
done
