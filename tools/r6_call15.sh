#!/bin/bash
# round 6, call 15: the three variants of call 14 b whose run stopped at a test that assumed the default pixel order / the kernels without the any-hit test; then the whole suite
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FH_VARIANTS="FH_PIXEL_BLOCK=0 FH_PIXEL_BLOCK=4 FH_FORCE_ALPHA=1" bash tools/gpu_variants.sh > gpurun_out/r06_variants_parity_c.log 2>&1; cat gpurun_out/r06_variants_parity_c.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_15_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_15_tests.log
