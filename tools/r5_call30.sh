cd $GRAFT_REPO_ROOT
# the GPU parity file under the round's switches (and a few of the older ones that interact with them)
: > gpurun_out/r05_variants_parity.log
for v in "FH_BOTTOM_UP=1 FH_STREAM=1" "FH_BOTTOM_UP=0 FH_STREAM=1" "FH_SAH_ITERS=4" "FH_SAH_ITERS=4 FH_BVH_BUILDER=ploc FH_STREAM=1" "FH_SUBPASS=3 FH_SUBPASS_MIN=1" "FH_OPACITY_MICROMAP=0" "FH_OPACITY_CLASSES=0" \
         "FH_PIPELINE=0 FH_BOTTOM_UP=1 FH_STREAM=1" "FH_STACK_LDS=1 FH_BOTTOM_UP=1 FH_STREAM=1" "FH_SORT=0 FH_BOTTOM_UP=1 FH_STREAM=1" "FH_MERGE=0" "FH_STREAM=0" "FH_COOP=0" "FH_BVH2=1"; do
  echo "== $v" >> gpurun_out/r05_variants_parity.log
  env $v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> gpurun_out/r05_variants_parity.log
done
cat gpurun_out/r05_variants_parity.log
