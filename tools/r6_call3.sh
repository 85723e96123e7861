#!/bin/bash
# round 6, call 3: where should the fused tail take over in a ONE-PASS call of the interior (configs[3])?  Survivors per bounce, then 1 / 4 / 16-spp call times with the switch
# depth forced (FH_TAIL_DEPTH) and with the survivor threshold varied (FH_TAIL_PATHS)
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_3_tail_sweep.log; : > $out
FH_DEBUG_TAIL=1 timeout -k 10 200 python tools/call_timeline.py run 3 1 3 2>&1 | grep "^\[tail\]" | tail -2 >> $out
FH_DEBUG_TAIL=1 timeout -k 10 200 python tools/call_timeline.py run 3 16 3 2>&1 | grep "^\[tail\]" | tail -2 >> $out
for v in "" FH_TAIL_DEPTH=2 FH_TAIL_DEPTH=3 FH_TAIL_DEPTH=4 FH_TAIL_DEPTH=5 FH_TAIL_DEPTH=6 FH_TAIL_DEPTH=8 FH_TAIL_PATHS=65536 FH_TAIL_PATHS=131072 FH_TAIL_PATHS=524288 FH_TAIL_PATHS=1048576; do
  echo "== ${v:-default}" >> $out
  env $v timeout -k 10 300 python tools/latency_small_calls.py 3 >> $out 2>&1 || { tail -3 $out; exit 1; }
done
cat $out
