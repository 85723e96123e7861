#!/bin/bash
# round 4, call 35: where the any-hit test's cost sits: the kernels with the test compiled in but alpha_pass accepting every candidate (variant stub: the image of FH_NO_ALPHA=1)
# against FH_NO_ALPHA=1 (kernels without the test) and the real thing; configs[3]
cd $GRAFT_REPO_ROOT
bash tools/gpu_ab.sh "stub2 stub" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
