cd $GRAFT_REPO_ROOT
for v in base old base old; do
  lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
  echo "== $v"; FH_LIB=$PWD/$lib timeout -k 10 300 python tools/latency_small_calls.py 3 2>&1 | grep configs
done > gpurun_out/r5_c20_latency.log 2>&1; cat gpurun_out/r5_c20_latency.log | cut -c1-200
