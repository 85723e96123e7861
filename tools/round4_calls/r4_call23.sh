#!/bin/bash
# round 4, call 23: 8-bit textures in 4 x 4 tiles (FH_TEX_TILED=0 = row-major as before): texture tests, then configs[3] off / on / off / on, FH_SORT=0 latency on [1] [2]
cd $GRAFT_REPO_ROOT
echo "== texture tests"; PYTHONFAULTHANDLER=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "texture or textured or random_materials or gltf or headless or garbage" > gpurun_out/r4_c23_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_c23_tests.log; [ $rc -eq 0 ] || exit 1
echo "== configs[3], 512 spp"; bash tools/gpu_env_ab.sh "FH_TEX_TILED=0 FH_TEX_TILED=1 FH_TEX_TILED=0 FH_TEX_TILED=1" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== latency configs[3]"; for v in FH_TEX_TILED=0 FH_TEX_TILED=1; do echo "-- $v"; env $v timeout -k 10 300 python3 tools/latency_breakdown.py 3 2>/dev/null | cut -c1-260; done
