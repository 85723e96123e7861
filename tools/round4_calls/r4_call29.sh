#!/bin/bash
# round 4, call 29: builder parameters on the 1 M-triangle soup (configs[2]): PLOC radius 16 (default) / 32 / 64, radix tree, greedy collapse; and the sky kernel in line in the serial step
cd $GRAFT_REPO_ROOT
bash tools/gpu_env_ab.sh "FH_X=0 FH_PLOC_RADIUS=32 FH_PLOC_RADIUS=64 FH_BVH_BUILDER=lbvh FH_BVH_BUILDER=ploc FH_COLLAPSE=greedy FH_SPLIT=0 FH_X=0" "2" "--steps 6 --warmup 2 --no-extras"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/env_FH_*_2.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    r=d["roofline"]; print(f.split("/")[-1], d["value"], d["bvh"], r.get("per_ray"))
PY
