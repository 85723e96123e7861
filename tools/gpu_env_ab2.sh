#!/bin/bash
# like gpu_env_ab.sh, but every variant is a quoted list of VAR=val pairs: tools/gpu_env_ab2.sh "<cfg>" "A=1 B=2" "A=2" ...
cd $GRAFT_REPO_ROOT
cfg=$1; shift
for v in "$@"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/env2_${tag}_$cfg.json 2> gpurun_out/env2_${tag}_$cfg.err || { echo "$v config $cfg FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/env2_${tag}_$cfg.json'))
a=d['kernel_ms_per_step_alone']
print('$v config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'tail', a['tail'], 'depth', d['bvh']['depth'], 'nodes', d['bvh']['nodes'])"
done
