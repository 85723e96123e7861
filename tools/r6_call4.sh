#!/bin/bash
# round 6, call 4: suite on the build with fh_unpack_shards / self-launching bench / frozen roofline fields / shade record; then the in-tile pixel order A/B (FH_PIXEL_BLOCK=8:
# the 64 lanes of a wave of k_generate hold an 8 x 8 patch instead of two rows of 32) on configs[3], [2] and [1]
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_4_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_4_tests.log
out=gpurun_out/r6_4_pixel_block.log; : > $out
for cfg in 3 2 1; do
  for v in "" FH_PIXEL_BLOCK=8 FH_PIXEL_BLOCK=4 ""; do
    spp=""; [ $cfg = 3 ] && spp="--spp 540"
    env $v timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[$cfg] ${v:-default}:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone', j['kernel_ms_per_step_alone'])" >> $out
  done
done
cat $out
