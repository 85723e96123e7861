#!/bin/bash
# round 4, call 26: k_sky_pixels as a background kernel: 1 / 2 / 3 / 4 workgroups per CU (grid-stride) instead of one thread per sky pixel, configs[2] and [4]
cd $GRAFT_REPO_ROOT
echo "== configs[2]"; bash tools/gpu_env_ab.sh "FH_SKY_BLOCKS=0 FH_SKY_BLOCKS=1 FH_SKY_BLOCKS=2 FH_SKY_BLOCKS=3 FH_SKY_BLOCKS=4 FH_SKY_BLOCKS=8 FH_SKY_BLOCKS=0" "2" "--steps 6 --warmup 2 --no-extras"
echo "== configs[4], 1024 spp"; bash tools/gpu_env_ab.sh "FH_SKY_BLOCKS=0 FH_SKY_BLOCKS=2 FH_SKY_BLOCKS=4" "4" "--spp 1024 --steps 2 --warmup 1 --no-extras"
