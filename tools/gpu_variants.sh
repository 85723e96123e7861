#!/bin/bash
# the GPU parity suite under every developer switch (results must not depend on any of them)
cd $GRAFT_REPO_ROOT
for v in ${FH_VARIANTS:-"FH_PIPELINE=0" "FH_OVERLAP=0" "FH_STREAM=0" "FH_COOP=0" "FH_BVH2=1" "FH_SORT=0" "FH_REFIT=0" "FH_BVH_BUILDER=ploc" "FH_SPLIT=0" "FH_PIPELINE=2" "FH_COLLAPSE=greedy" "FH_STREAM=1" "FH_ABSORB=0" "FH_STREAM_MIN_RAYS=0" "FH_STREAM_MIN_RAYS=4096" "FH_SORT_SMALL=1" "FH_TAIL_PATHS=1024" "FH_TAIL_DEPTH=2" "FH_STREAM_REFILL=8" "FH_COOP_T=8" "FH_MERGE=0" "FH_SHADE_STREAM=2" "FH_STACK_LDS=3" "FH_STACK_LDS=99"}; do
  env $v python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pins.py -m gpu -x -q > gpurun_out/variant_${v%%=*}_${v##*=}.log 2>&1
  echo "$v rc=$? $(tail -1 gpurun_out/variant_${v%%=*}_${v##*=}.log)"
done
