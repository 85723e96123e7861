#!/bin/bash
# round 4, call 8: queue appends of k_generate / k_shade with one returning atomic per workgroup (base) against one per wave (variant wavepush), then the full GPU suite
cd $GRAFT_REPO_ROOT
echo "== configs[3], 512 spp"; bash tools/gpu_ab.sh "wavepush base wavepush base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== configs[1]"; bash tools/gpu_ab.sh "wavepush base wavepush base" "1" "--steps 4 --warmup 1 --no-extras"
echo "== configs[2]"; bash tools/gpu_ab.sh "wavepush base wavepush base" "2" "--steps 6 --warmup 2 --no-extras"
echo "== full GPU suite"; timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
