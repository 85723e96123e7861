#!/bin/bash
# the GPU parity suite under every developer switch (results must not depend on any of them)
cd $GRAFT_REPO_ROOT
for v in ${FH_VARIANTS:-"FH_PIPELINE=0" "FH_OVERLAP=0" "FH_STREAM=0" "FH_COOP=0" "FH_BVH2=1" "FH_SORT=0" "FH_REFIT=0" "FH_BVH_BUILDER=ploc" "FH_SPLIT=0" "FH_PIPELINE=2" "FH_COLLAPSE=greedy" "FH_STREAM=1" "FH_ABSORB=0" "FH_STREAM_MIN_RAYS=0" "FH_STREAM_MIN_RAYS=4096" "FH_SORT_SMALL=1" "FH_TAIL_PATHS=1024" "FH_TAIL_DEPTH=2" "FH_STREAM_REFILL=8" "FH_COOP_T=8" "FH_MERGE=0" "FH_SORT_ONEPASS=0" "FH_SORT_ONEPASS=1" "FH_PIXEL_BLOCK=0" "FH_PIXEL_BLOCK=4" "FH_SHADE_WGS=2" "FH_FORCE_ALPHA=1" "FH_STREAM_CHUNK=16" "FH_BOTTOM_UP=1" "FH_STACK_LDS=3" "FH_STACK_LDS=99" "FH_POISON=1" "FH_SKY_SPLIT_MIN_LOG2=0" "FH_SKY_SPLIT=0" "FH_SKY_BLOCKS=1" "FH_SKY_PRIO=0"}; do
  env $v python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pins.py -m gpu -x -q > gpurun_out/variant_${v%%=*}_${v##*=}.log 2>&1
  rc=$?
  echo "$v rc=$rc $(tail -1 gpurun_out/variant_${v%%=*}_${v##*=}.log)"
  # a run that was killed, aborted or reported a GPU fault is the last GPU step of this call
  if [ $rc -ge 124 ] || grep -q "Memory access fault\|Fatal Python error" gpurun_out/variant_${v%%=*}_${v##*=}.log; then echo "stopping after $v"; exit 1; fi
done
