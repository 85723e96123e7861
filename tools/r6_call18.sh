#!/bin/bash
# round 6, call 18: robustness of the final build -- soak (the same 1080p frame again and again, every one bit-identical: the soup, and the interior with its cut-outs and spilled
# stack), the randomized material / texture / light parity test over 72 more seeds, the sky-pixel split under 200 random cameras
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r06_soak_fuzz.log; : > $out
SOAK_FRAMES=100 timeout -k 10 300 python tools/soak.py 2>&1 | tail -3 >> $out
SCENE=sponza SOAK_FRAMES=100 timeout -k 10 400 python tools/soak.py 2>&1 | tail -3 >> $out
timeout -k 10 600 python tools/fuzz_more.py 9 81 2>&1 | tail -4 >> $out
timeout -k 10 300 python tools/fuzz_sky_split.py 200 6 2>&1 | tail -3 >> $out
cat $out
