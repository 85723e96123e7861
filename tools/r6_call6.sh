#!/bin/bash
# round 6, call 6: what a wavefront bounce of FEW paths costs, launch by launch: the 1-spp timeline of configs[3] with the fused tail switched off (FH_TAIL_DEPTH=8: all eight
# bounces through the wavefront kernels; 142 k / 64 k / 30 k / 14 k / 7 k paths in bounces 3 .. 7), and with the cell sorts off as well
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "FH_TAIL_DEPTH=8" "FH_TAIL_DEPTH=8 FH_SORT=0"; do
  tag=$(echo $v | tr ' =' '__')
  rm -rf gpurun_out/tl_x
  env $v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_x -o tl -- python3 tools/call_timeline.py run 3 1 40 > gpurun_out/r6_6_run_$tag.log 2>&1 || { tail -5 gpurun_out/r6_6_run_$tag.log; exit 1; }
  grep "configs\[" gpurun_out/r6_6_run_$tag.log
  f=$(find gpurun_out/tl_x -name "*kernel_trace.csv" | head -1)
  python3 tools/call_timeline.py reduce $f 1 40 > gpurun_out/r6_timeline_config3_1spp_$tag.txt && tail -4 gpurun_out/r6_timeline_config3_1spp_$tag.txt
  rm -rf gpurun_out/tl_x
done
