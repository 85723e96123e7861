#!/bin/bash
# round 4, call 22: what a 1-spp / 16-spp frame of configs[3] (and [2]) does under the switches that shape small launches: tail hand-over, queue sort, workgroups taking part, merge
cd $GRAFT_REPO_ROOT
for v in "FH_X=0" "FH_TAIL_PATHS=32768" "FH_TAIL_PATHS=65536" "FH_TAIL_PATHS=131072" "FH_TAIL_PATHS=524288" "FH_TAIL_PATHS=1048576" "FH_TAIL_DEPTH=8" "FH_TAIL_DEPTH=2" "FH_TAIL_DEPTH=1" "FH_SORT=0" "FH_STREAM_MIN_RAYS=0" "FH_STREAM_MIN_RAYS=32" "FH_STREAM_MIN_RAYS=128" "FH_STREAM_MIN_RAYS=256" "FH_MERGE=0" "FH_STREAM_CHUNK=64" "FH_X=0"; do
  echo "== $v"
  env $v timeout -k 10 300 python3 tools/latency_breakdown.py 3 2 2> gpurun_out/lat.err | cut -c1-330 || { echo FAILED; tail -3 gpurun_out/lat.err; exit 1; }
done 2>&1 | tee gpurun_out/r4_c22_latency_sweep.txt
