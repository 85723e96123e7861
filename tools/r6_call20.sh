#!/bin/bash
# round 6, call 20: the two tunables of the streaming traversal once more on this round's build (refill threshold 24, flush threshold 48): configs[3] (540 spp) and configs[2]
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_20_tunables.log; : > $out
for cfg in 3 2; do
  for v in "" FH_STREAM_REFILL=16 FH_STREAM_REFILL=32 FH_COOP_T=40 FH_COOP_T=56 FH_STREAM_CHUNK_CLOSEST=96 FH_STREAM_CHUNK_CLOSEST=192 ""; do
    spp=""; [ $cfg = 3 ] && spp="--spp 540"
    env $v timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg] ${v:-default}:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'shade', a['shade'])" >> $out
  done
done
cat $out
