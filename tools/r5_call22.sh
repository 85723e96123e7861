cd $GRAFT_REPO_ROOT
for v in base old base old; do
  lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
  echo "== $v"; FH_LIB=$PWD/$lib timeout -k 10 300 python tools/latency_small_calls.py 3 2 1 2>&1 | grep configs
done > gpurun_out/r5_c22.log 2>&1; cut -c1-200 gpurun_out/r5_c22.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_c22_tests.log 2>&1; tail -2 gpurun_out/r5_c22_tests.log
bash tools/gpu_env_ab3.sh "FH_BOTTOM_UP=0" "2" "--steps 8" 2>&1 | cut -c1-200
