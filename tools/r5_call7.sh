cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "FH_SUBPASS=1" "FH_SUBPASS=3" "FH_SUBPASS=2"; do env $v timeout -k 10 300 python tools/latency_small_calls.py 2 3 1 2>&1 | grep configs; done; done > gpurun_out/r5_c7_latency.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_c7_tests.log 2>&1; tail -3 gpurun_out/r5_c7_tests.log
