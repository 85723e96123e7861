#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the bench configurations: tools/gpu_ab.sh "<variant names, '' = the default build>" "<configs>"
cd $GRAFT_REPO_ROOT
for cfg in $2; do
  for v in $1; do
    lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
    FH_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --config $cfg --no-cpu-baseline $3 > gpurun_out/ab_${v}_$cfg.json 2> gpurun_out/ab_${v}_$cfg.err || { echo "$v config $cfg FAILED"; tail -3 gpurun_out/ab_${v}_$cfg.err; continue; }
    python3 -c "
import json,sys
d=json.load(open('gpurun_out/ab_${v}_$cfg.json'))
a=d['kernel_ms_per_step_alone']
print('$v config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'generate', a['generate'], 'tail', a['tail'])"
  done
done
