#!/bin/bash
# round 6, call 24: the fused tail streaming its wave's rays (TailStream) against rounds of 64 (FH_TAIL_STREAM=0): parity first (the tail-heavy settings too), then the
# one-pass calls of configs[3], [2], [1] and the full frames
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_24_tail_stream.log; : > $out
for v in "" "FH_TAIL_PATHS=1024" "FH_TAIL_DEPTH=1" "FH_TAIL_DEPTH=2 FH_STREAM=1"; do
  env $v timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_24_tests.log 2>&1; rc=$?
  echo "parity ${v:-default}: rc $rc $(tail -1 gpurun_out/r6_24_tests.log)" >> $out
  [ $rc -eq 0 ] || { cat $out; grep -n "FAILED\|Error" gpurun_out/r6_24_tests.log | head -5; exit 1; }
done
for v in "" FH_TAIL_STREAM=0 "" FH_TAIL_STREAM=0; do
  echo "== ${v:-streamed (default)}" >> $out
  env $v timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1
done
for cfg in 2 3; do
  for v in "" FH_TAIL_STREAM=0; do
    spp=""; [ $cfg = 3 ] && spp="--spp 540"
    env $v timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg] ${v:-streamed}:', j['value'], 'Msamples/s; alone tail', a['tail'], 'total', a['render_total'])" >> $out
  done
done
cat $out
