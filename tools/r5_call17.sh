cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5_c17_tests.log 2>&1; tail -3 gpurun_out/r5_c17_tests.log
bash tools/gpu_env_ab3.sh "FH_BOTTOM_UP=0" "2" "--steps 8" > gpurun_out/r5_c17_ab.log 2>&1
cut -c1-250 gpurun_out/r5_c17_ab.log
