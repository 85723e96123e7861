#!/bin/bash
# round 4, call 37: k_generate with eight chunks of 256 pixels per queue atomic (four before): sampler / camera / split tests, then the three configurations (compare with the `base`
# rows of call 36: generate alone 50.4 / 13.4 / 36.7 ms)
cd $GRAFT_REPO_ROOT
echo "== tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "camera or tiny or small_path_pool or sky_pixel or cornell or garbage" > gpurun_out/r4_c37_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c37_tests.log; [ $rc -eq 0 ] || exit 1
for cfg in 2 1 3; do
  extra="--steps 6 --warmup 2"; [ $cfg = 3 ] && extra="--spp 512 --steps 2 --warmup 1"
  for rep in 1 2; do
  timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $extra > gpurun_out/c37_$cfg.json 2> gpurun_out/c37_$cfg.err || { echo FAILED; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/c37_$cfg.json')); a=d['kernel_ms_per_step_alone']
print('config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone route+sort', a['route_and_sort'], 'shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'generate', a['generate'])"
  done
done
