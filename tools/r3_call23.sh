#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { cfg=$1; shift; env "$@" timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras --steps $STEPS --warmup 1 > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$* FAILED"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); a=d['kernel_ms_per_step_alone']
print('config $cfg $*', d['value'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'tail', a['tail'], 'shade', a['shade'])"; }
STEPS=4
run 2 FH_X=0
for c in 96 128 192 256 512; do run 2 FH_STREAM_CHUNK_CLOSEST=$c; done
run 2 FH_STREAM_CHUNK_CLOSEST=128 FH_STREAM_WGS=5
run 2 FH_X=0
STEPS=1
run 3 FH_X=0
run 3 FH_STREAM_CHUNK_CLOSEST=128
run 4 FH_X=0
run 4 FH_STREAM_CHUNK_CLOSEST=128
