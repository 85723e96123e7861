#!/bin/bash
# A/B of environment switches on the bench configurations: tools/gpu_env_ab.sh "<VAR=val ...>" "<configs>" [extra bench args]
cd $GRAFT_REPO_ROOT
for cfg in $2; do
  for v in $1; do
    env $v timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline $3 > gpurun_out/env_${v}_$cfg.json 2> gpurun_out/env_${v}_$cfg.err || { echo "$v config $cfg FAILED"; tail -3 gpurun_out/env_${v}_$cfg.err; continue; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/env_${v}_$cfg.json'))
a=d['kernel_ms_per_step_alone']
print('$v config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'generate', a['generate'], 'tail', a['tail'], 'sort', a['route_and_sort'])"
  done
done
