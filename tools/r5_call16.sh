cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "start_at_the_node" > gpurun_out/r5_c16_tests.log 2>&1; tail -3 gpurun_out/r5_c16_tests.log
bash tools/gpu_env_ab3.sh "FH_BOTTOM_UP=0" "2" "--steps 8" > gpurun_out/r5_c16_ab.log 2>&1
bash tools/gpu_env_ab3.sh "FH_BOTTOM_UP=0" "4" "--spp 2048 --steps 1" >> gpurun_out/r5_c16_ab.log 2>&1
cut -c1-250 gpurun_out/r5_c16_ab.log
