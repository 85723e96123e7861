#!/bin/bash
# round 4, call 36: face ids of the closest hits per queue ENTRY (PoolDev::q_prim): k_route and k_miss_primary read two streams instead of a path record per ray.
# GPU suite, then configs[2] / [1] / [3] against the library of the commit before (prev)
cd $GRAFT_REPO_ROOT
echo "== GPU suite"; PYTHONFAULTHANDLER=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_c36_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c36_tests.log; [ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4_c36_tests.log | head; exit 1; }
ab() { cfg=$1; extra=$2; shift 2; for v in "$@"; do lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
  FH_LIB=$PWD/$lib timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $extra > gpurun_out/ab_${v}_$cfg.json 2> gpurun_out/ab_${v}_$cfg.err || { echo "$v FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/ab_${v}_$cfg.json')); a=d['kernel_ms_per_step_alone']
print('$v config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone route+sort', a['route_and_sort'], 'shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'generate', a['generate'])"; done; }
echo "== configs[2]"; ab 2 "--steps 6 --warmup 2" prev base prev base
echo "== configs[1]"; ab 1 "--steps 6 --warmup 2" prev base prev base
echo "== configs[3]"; ab 3 "--spp 512 --steps 2 --warmup 1" prev base
