#!/bin/bash
# round 6, call 23: how long is the fused tail as a function of the paths it takes over?  1-spp timelines of configs[3] with the hand-over forced to depth 3 (142 k paths, five
# bounces), 4 (64 k, four), 5 (30 k, three), 6 (14 k, two)
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_tail_vs_paths.log; : > $out
for d in 3 4 5 6; do
  rm -rf gpurun_out/tl_x
  FH_TAIL_DEPTH=$d timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_x -o tl -- python3 tools/call_timeline.py run 3 1 30 > gpurun_out/r6_23_run_$d.log 2>&1 || { tail -5 gpurun_out/r6_23_run_$d.log; exit 1; }
  f=$(find gpurun_out/tl_x -name "*kernel_trace.csv" | head -1)
  python3 tools/call_timeline.py reduce $f 1 30 > gpurun_out/r6_23_tl_$d.txt
  echo "FH_TAIL_DEPTH=$d: $(grep 'k_tail' gpurun_out/r6_23_tl_$d.txt | awk '{print "k_tail " $5 " us"}'); $(grep '^# span' gpurun_out/r6_23_tl_$d.txt)" >> $out
  rm -rf gpurun_out/tl_x
done
cat $out
