cd $GRAFT_REPO_ROOT
# which of r5-12's two defaults costs the 1-spp call of configs[2] its 0.17 ms?  fh_render(n) + fh_sync, n = 1 / 4 / 16, median / min (tools/latency_small_calls.py)
for v in "FH_X=0" "FH_COOP_T=32" "FH_SHADE_WGS=2" "FH_COOP_T=32 FH_SHADE_WGS=2" "FH_X=0"; do
  echo "== $v"; env $v timeout -k 10 200 python tools/latency_small_calls.py 2 3 2>/dev/null | cut -c1-220 || exit 1
done > gpurun_out/r5_latency_defaults.log 2>&1; rc=$?; cat gpurun_out/r5_latency_defaults.log; exit $rc
