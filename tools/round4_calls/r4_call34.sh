#!/bin/bash
# round 4, call 34: alpha_pass with global texel loads and the alpha texture's record entry read only where a face has one (default build) against the build before it (ring) and the
# commit before the ring (prev); configs[3]
cd $GRAFT_REPO_ROOT
echo "== tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "texture or textured or random_materials or gltf or headless or any_hit or alpha" > gpurun_out/r4_c34_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c34_tests.log; [ $rc -eq 0 ] || { grep -n "Error\|assert" gpurun_out/r4_c34_tests.log | head; exit 1; }
echo "== configs[3]"; bash tools/gpu_ab.sh "ring base prev ring base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
