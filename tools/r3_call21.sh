#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FH_BVH2=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_pins.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/gpu_ab.sh "base" "2" "--steps 4 --warmup 1 --no-extras"
