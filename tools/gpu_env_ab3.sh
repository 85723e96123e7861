#!/bin/bash
# same-box A/B of an environment switch: tools/gpu_env_ab3.sh "<VAR=value>" "<configs>" "<bench args>"   (each configuration: off, on, off, on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in $2; do
  for rep in 1 2; do
    for v in "FH_AB_OFF=1" "$1"; do
      env $v timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v config $cfg FAILED"; tail -2 gpurun_out/ab.err; continue; }
      python3 -c "
import json
d=json.load(open('gpurun_out/ab.json')); a=d['kernel_ms_per_step_alone']
print('config $cfg $v:', d['value'], 'Msamples/s; alone shade', a['shade'], 'route+sort', a['route_and_sort'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'])"
    done
  done
done
