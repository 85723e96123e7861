cd $GRAFT_REPO_ROOT
for rep in 1 2; do timeout -k 10 300 python tools/latency_small_calls.py 3 2 2>&1 | grep configs; done > gpurun_out/r5_c19_latency.log 2>&1; cat gpurun_out/r5_c19_latency.log
FH_DEBUG_BVH=1 SPP=256 VARIANTS="FH_BOTTOM_UP=2" timeout -k 10 900 python tools/sah_compare.py soup 2>&1 | grep "start at\|^soup" | cut -c1-200
