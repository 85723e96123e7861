cd $GRAFT_REPO_ROOT
FH_BOTTOM_UP=1 FH_STREAM=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5_c15_tests_forced.log 2>&1; tail -3 gpurun_out/r5_c15_tests_forced.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r5_c15_tests.log 2>&1; tail -3 gpurun_out/r5_c15_tests.log
SPP=256 VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=2;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py soup soup4 > gpurun_out/r5_bu6.log 2>&1
SPP=64 VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=2;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py city >> gpurun_out/r5_bu6.log 2>&1
SPP=64 VARIANTS="FH_BOTTOM_UP=0,FH_NO_ALPHA=1;FH_BOTTOM_UP=2,FH_NO_ALPHA=1" timeout -k 10 900 python tools/sah_compare.py sponza >> gpurun_out/r5_bu6.log 2>&1
grep "^soup\|^city\|^sponza\|start at" gpurun_out/r5_bu6.log | sed 's/FH_SAH_ITERS=default  builder=auto : build [0-9. ms(call)]*, //; s/, wave steps.*//' | cut -c1-330
