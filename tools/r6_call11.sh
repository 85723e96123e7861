#!/bin/bash
# round 6, call 11: the GPU suite on the pruned build (switches at their defaults removed: the device code is the same but for one LDS store), bench of configs 2 / 3 / 1
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_11_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r6_11_tests.log
[ $rc -eq 0 ] || exit 1
for cfg in 2 3 1; do
  spp=""; [ $cfg = 3 ] && spp="--spp 540"
  timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg]:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'shade', a['shade'], 'total', a['render_total'])"
done
