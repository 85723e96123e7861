#!/bin/bash
# round 6, call 29 (a / b): the GPU parity suite under every developer switch, on the round's final build (two halves: a call is at most 20 minutes)
cd $GRAFT_REPO_ROOT
if [ "$1" = a ]; then
  export FH_VARIANTS='FH_PIPELINE=0 FH_OVERLAP=0 FH_STREAM=0 FH_COOP=0 FH_BVH2=1 FH_SORT=0 FH_REFIT=0 FH_BVH_BUILDER=ploc FH_SPLIT=0 FH_PIPELINE=2 FH_COLLAPSE=greedy FH_STREAM=1 FH_ABSORB=0 FH_STREAM_MIN_RAYS=0 FH_STREAM_MIN_RAYS=4096 FH_SORT_SMALL=1 FH_TAIL_PATHS=1024 FH_TAIL_DEPTH=2'
else
  export FH_VARIANTS='FH_STREAM_REFILL=8 FH_COOP_T=8 FH_MERGE=0 FH_SORT_ONEPASS=0 FH_SORT_ONEPASS=1 FH_PIXEL_BLOCK=0 FH_PIXEL_BLOCK=4 FH_SHADE_WGS=2 FH_FORCE_ALPHA=1 FH_STREAM_CHUNK=16 FH_BOTTOM_UP=1 FH_STACK_LDS=3 FH_STACK_LDS=99 FH_POISON=1 FH_SKY_SPLIT_MIN_LOG2=0 FH_SKY_SPLIT=0 FH_SKY_BLOCKS=1 FH_SKY_PRIO=0'
fi
bash tools/gpu_variants.sh > gpurun_out/r06_variants_$1.log 2>&1; rc=$?
cat gpurun_out/r06_variants_$1.log
exit $rc
