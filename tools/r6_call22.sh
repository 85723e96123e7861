#!/bin/bash
# round 6, call 22: k_sky_pixels without the AOV running means of pixels whose AOVs are exactly +0 (they stay +0): configs[2] and [4], new build against the build before
# (fredholm_amd/libfredholm_hip_base.so), the sky-split parity tests and the split fuzz on the new build
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_22_sky_aov.log; : > $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "sky or config or full_size or progressive or pool" -p no:cacheprovider 2>&1 | tail -2 >> $out
timeout -k 10 300 python tools/fuzz_sky_split.py 200 8 2>&1 | tail -1 >> $out
for cfg in 2 4; do
  for v in base new base new; do
    lib=fredholm_amd/libfredholm_hip.so; [ "$v" = base ] && lib=fredholm_amd/libfredholm_hip_base.so
    spp=""; [ $cfg = 4 ] && spp="--spp 1024 --steps 3 --warmup 1"
    FH_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg] $v:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone generate+sky', a['generate'], 'total', a['render_total'])" >> $out
  done
done
cat $out
