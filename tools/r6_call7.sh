#!/bin/bash
# round 6, call 7: do the cell sorts of the bounce queues pay in ONE-PASS calls?  fh_render(1 / 4 / 16) + fh_sync with and without them (FH_SORT=0), configs[3], [2], [1]
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_7_sort_small_calls.log; : > $out
for v in "" FH_SORT=0 ""; do
  echo "== ${v:-default}" >> $out
  env $v timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1 || { tail -3 $out; exit 1; }
done
cat $out
