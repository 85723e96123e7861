#!/bin/bash
# A/B of bench arguments: tools/gpu_args_ab.sh <config> "<args variant 1>" "<args variant 2>" ...
cd $GRAFT_REPO_ROOT
cfg=$1; shift
for v in "$@"; do
  tag=$(echo "$v" | tr ' =-' '___')
  timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline $v > gpurun_out/args_${tag}_$cfg.json 2> gpurun_out/args_${tag}_$cfg.err || { echo "$v config $cfg FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/args_${tag}_$cfg.json'))
print('[$v] config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', d['step_ms'])"
done
