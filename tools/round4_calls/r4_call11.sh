#!/bin/bash
# round 4, call 11: sky-pixel split: its own test, the parity suite with the split forced on every call, bench A/B (FH_SKY_SPLIT=0 = every pixel through the passes), and the
# unordered any-hit variant
cd $GRAFT_REPO_ROOT
echo "== split test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sky_pixel_split" 2>&1 | tail -5
echo "== parity suite, split forced"; FH_SKY_SPLIT_MIN_LOG2=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
echo "== configs[2]"; bash tools/gpu_env_ab.sh "FH_SKY_SPLIT=0 FH_X=0 FH_SKY_SPLIT=0 FH_X=0" "2" "--steps 6 --warmup 2 --no-extras"
echo "== configs[4], 1024 spp"; bash tools/gpu_env_ab.sh "FH_SKY_SPLIT=0 FH_X=0" "4" "--spp 1024 --steps 2 --warmup 1 --no-extras"
echo "== configs[3] (no sky pixels: must be unchanged)"; bash tools/gpu_env_ab.sh "FH_SKY_SPLIT=0 FH_X=0" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== unordered any-hit"; bash tools/gpu_ab.sh "base unord base unord" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"; bash tools/gpu_ab.sh "base unord" "2" "--steps 6 --warmup 2 --no-extras"
