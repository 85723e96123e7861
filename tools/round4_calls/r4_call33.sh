#!/bin/bash
# round 4, call 33: parked any-hit tests in the closest-hit launch only (default build), with the ring worked off at 8 entries (f8), and in every streaming kernel at 4 (mix4),
# against the library of the commit before (prev); configs[3]
cd $GRAFT_REPO_ROOT
echo "== tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "texture or textured or random_materials or gltf or headless or config3 or sponza or any_hit or alpha or closest" > gpurun_out/r4_c33_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c33_tests.log; [ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4_c33_tests.log | head; exit 1; }
echo "== configs[3]"; bash tools/gpu_ab.sh "prev base f8 mix4 prev base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
