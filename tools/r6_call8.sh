#!/bin/bash
# round 6, call 8: one-pass calls with RAY items in the secondary launches (SecondaryStream<.., RAYS> + k_resolve_secondary) and without the cell sorts that buy them nothing:
# parity suite first (the new forms are the defaults), then fh_render(1 / 4 / 16) + fh_sync under each switch, configs[3], [2], [1]
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_8_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r6_8_tests.log
[ $rc -eq 0 ] || exit 1
out=gpurun_out/r6_8_ray_items.log; : > $out
for v in "" "FH_RAY_ITEMS=0 FH_SORT_ONEPASS=1" "FH_RAY_ITEMS=0" "FH_SORT_ONEPASS=1" "FH_SORT_ONEPASS=0" "FH_RAY_ITEMS_DEPTH=1" "FH_RAY_ITEMS_DEPTH=2" "FH_TAIL_DEPTH=4" "FH_TAIL_DEPTH=5" "FH_TAIL_DEPTH=8" "FH_TAIL_PATHS=131072" "FH_TAIL_PATHS=65536" ""; do
  echo "== ${v:-default}" >> $out
  env $v timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1 || { tail -3 $out; exit 1; }
done
cat $out
