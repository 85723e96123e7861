cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "alpha or cut or textur or sponza or ring" > gpurun_out/r5_c6_tests.log 2>&1 || { tail -30 gpurun_out/r5_c6_tests.log; exit 1; }
bash tools/gpu_ab.sh "base nodefer nosusp base nodefer" "3" "--spp 512 --steps 2 --no-extras" > gpurun_out/r5_c6_ab.log 2>&1
