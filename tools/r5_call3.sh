cd $GRAFT_REPO_ROOT
VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py soup sponza city > gpurun_out/r5_bu1.log 2>&1 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5_bu1_tests.log 2>&1
