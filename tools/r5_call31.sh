cd $GRAFT_REPO_ROOT
: > gpurun_out/r05_variants_parity_b.log
for v in "FH_OPACITY_MICROMAP=0" "FH_OPACITY_CLASSES=0"; do
  echo "== $v (after the opacity test was made to clear the switches it is about)" >> gpurun_out/r05_variants_parity_b.log
  env $v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> gpurun_out/r05_variants_parity_b.log
done
cat gpurun_out/r05_variants_parity_b.log
