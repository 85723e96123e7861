#!/bin/bash
# round 6, calls 14 a / b: the GPU parity files under every developer switch that is left (tools/gpu_variants.sh), in two halves
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ "$1" = a ]; then  # (second issue of this half: the three variants whose run stopped at a test of this round that assumed the default kernels)
  export FH_VARIANTS="FH_PIPELINE=0 FH_COOP=0 FH_BVH2=1"
else
  export FH_VARIANTS="FH_STREAM_REFILL=8 FH_COOP_T=8 FH_MERGE=0 FH_SORT_ONEPASS=0 FH_SORT_ONEPASS=1 FH_PIXEL_BLOCK=0 FH_PIXEL_BLOCK=4 FH_SHADE_WGS=2 FH_FORCE_ALPHA=1 FH_STREAM_CHUNK=16 FH_BOTTOM_UP=1 FH_STACK_LDS=3 FH_STACK_LDS=99 FH_POISON=1 FH_SKY_SPLIT_MIN_LOG2=0 FH_SKY_SPLIT=0 FH_SKY_BLOCKS=1 FH_SKY_PRIO=0"
fi
bash tools/gpu_variants.sh > gpurun_out/r06_variants_parity_$1.log 2>&1
cat gpurun_out/r06_variants_parity_$1.log
