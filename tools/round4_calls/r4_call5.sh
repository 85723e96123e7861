#!/bin/bash
# round 4, call 5: node fetch by pairs of lanes in the streaming kernels (7 and 6 workgroups per CU) against the per-lane fetch, parity suite under it
cd $GRAFT_REPO_ROOT
echo "== parity suite with the pair fetch"; FH_LIB=$PWD/fredholm_amd/libfredholm_hip_p7.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== configs[3], 512 spp"; bash tools/gpu_ab.sh "base p7 p6 base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== configs[2]"; bash tools/gpu_ab.sh "base p7 p6 base" "2" "--steps 6 --warmup 2 --no-extras"
