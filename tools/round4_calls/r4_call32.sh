#!/bin/bash
# round 4, call 32: any-hit tests parked in a per-wave ring and worked off together (fh_trace.h: alpha_ring / alpha_flush): the tests of textured and cut-out scenes, then
# configs[3] against the library of the commit before (variant "prev"), twice each
cd $GRAFT_REPO_ROOT
echo "== tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "texture or textured or random_materials or gltf or headless or garbage or config3 or sponza or any_hit or alpha or closest" > gpurun_out/r4_c32_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c32_tests.log; [ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4_c32_tests.log | head; exit 1; }
echo "== configs[3]"; bash tools/gpu_ab.sh "prev base prev base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
