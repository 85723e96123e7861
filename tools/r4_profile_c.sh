#!/bin/bash
# round 4, profile call C: the round's profile set of one more configuration; the counter passes render three passes of the size the configuration's default run uses
# usage: bash tools/r4_profile_c.sh <config> <tag>
cd $GRAFT_REPO_ROOT
cfg=$1; tag=$2
spp=$(python3 bench.py --config $cfg --no-cpu-baseline --no-extras --steps 1 --warmup 0 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['spp_per_pass'])") || exit 1
echo "spp per pass $spp"; echo $((3 * spp)) > gpurun_out/${tag}_pspp.txt
bash tools/profile_round3.sh $tag $cfg $((3 * spp))
