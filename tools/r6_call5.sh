#!/bin/bash
# round 6, call 5: suite on the default build (8 x 8 pixel blocks by default now), then same-box A/B of three library variants on configs[3] (540 spp), [2] and [1]:
#   rec8     face records padded to one 128-byte line (FH_FACE_REC_STRIDE=8)
#   clsprim  the face id of every hit next to its class-queue entry: k_shade asks for the face record without waiting for the hit record (FH_CLS_PRIM=1)
#   both
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_5_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_5_tests.log
out=gpurun_out/r6_5_variants.log; : > $out
for v in rec8 clsprim both; do  # the variants must render the same bits: smoke() against the checker with each
  FH_LIB=$PWD/fredholm_amd/libfredholm_hip_$v.so timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1 || { echo "variant $v: smoke FAILED" >> $out; }
done
for cfg in 3 2 1; do
  for v in base rec8 clsprim both base; do
    lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
    spp=""; [ $cfg = 3 ] && spp="--spp 540"
    FH_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg] $v:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone shade', a['shade'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'route', a['route_and_sort'], 'total', a['render_total'])" >> $out
  done
done
cat $out
