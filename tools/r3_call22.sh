#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$* FAILED"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); a=d['kernel_ms_per_step_alone']
print('$*', d['value'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'tail', a['tail'], 'shade', a['shade'])"; }
run FH_COOP_T=32
for t in 16 24 48; do run FH_COOP_T=$t; done
for w in 5 4; do run FH_STREAM_WGS=$w; done
for tp in 32768 16384; do run FH_TAIL_PATHS=$tp; done
for c in 128 32; do run FH_STREAM_CHUNK=$c; done
run FH_COOP_T=32
