cd $GRAFT_REPO_ROOT
for v in base old; do
  lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
  echo "== $v"; FH_LIB=$PWD/$lib timeout -k 10 300 python tools/latency_breakdown.py 3 2>&1 | grep "config 3"
done > gpurun_out/r5_c21.log 2>&1; cut -c1-330 gpurun_out/r5_c21.log
