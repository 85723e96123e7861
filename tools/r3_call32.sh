#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tp in 262144 524288; do echo "FH_TAIL_PATHS=$tp"; FH_TAIL_PATHS=$tp timeout -k 10 300 python3 tools/latency_breakdown.py 1 2 2>&1 | cut -c1-130; done
run() { cfg=$1; shift; env "$@" timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras --steps $STEPS --warmup 1 > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$* FAILED"; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/sw.json')); a=d['kernel_ms_per_step_alone']
print('config $cfg $*', d['value'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'tail', a['tail'], 'shade', a['shade'])"; }
STEPS=4
run 2 FH_X=0; run 2 FH_TAIL_PATHS=131072; run 2 FH_TAIL_PATHS=262144; run 2 FH_X=0
STEPS=2
run 4 FH_X=0; run 4 FH_TAIL_PATHS=131072; run 4 FH_TAIL_PATHS=262144; run 4 FH_TAIL_PATHS=524288
run 1 FH_X=0; run 1 FH_TAIL_PATHS=262144; run 1 FH_TAIL_PATHS=1048576
