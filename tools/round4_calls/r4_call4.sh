#!/bin/bash
# round 4, call 4: same-box matrix of {6, 7 workgroups per CU} x {scan hand-over on / off} (+ the any-hit kernels at 6, + the scan only without the any-hit test)
cd $GRAFT_REPO_ROOT
echo "== configs[3], 512 spp"; bash tools/gpu_ab.sh "old w6s1 w7s0 base sa0 a6 old" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== configs[2]"; bash tools/gpu_ab.sh "old w6s1 w7s0 base old" "2" "--steps 6 --warmup 2 --no-extras"
