cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_c11_tests.log 2>&1; tail -3 gpurun_out/r5_c11_tests.log
bash tools/gpu_ab.sh "notop base notop base" "2" "--steps 8 --no-extras" > gpurun_out/r5_c11_ab.log 2>&1
bash tools/gpu_ab.sh "notop base notop base" "3" "--spp 512 --steps 2 --no-extras" >> gpurun_out/r5_c11_ab.log 2>&1
cat gpurun_out/r5_c11_ab.log
