#!/bin/bash
# round 4, call 1: same-box A/B of the occupancy variants (7 / 8 workgroups per CU), the scan hand-over, BVH build parameters on configs[3], and where a 1-spp configs[3] frame goes
cd $GRAFT_REPO_ROOT
echo "== variants on configs[3], 512 spp"; bash tools/gpu_ab.sh "base wg7 wg8 scan base" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== variants on configs[2]"; bash tools/gpu_ab.sh "base wg7 wg8 scan" "2" "--steps 6 --warmup 2 --no-extras"
echo "== BVH parameters on configs[3] (sponza_probe 128 spp)"
for v in "FH_DEBUG_BVH=1" "FH_PLOC_RADIUS=32" "FH_PLOC_RADIUS=64" "FH_BVH_BUILDER=lbvh" "FH_SPLIT_DIV=64 FH_SPLIT_BUDGET=2" "FH_SPLIT=0"; do
  echo "-- $v"; env FH_DEBUG_BVH=1 $v timeout -k 10 200 python3 tools/sponza_probe.py 128 2>&1 | grep -v "^\[trace\]\|^\[tail\]" | cut -c1-400
done
echo "== latency breakdown configs[3]"; timeout -k 10 200 python3 tools/latency_breakdown.py 3 2>&1 | tail -4
