#!/bin/bash
# round 4, call 21: state after the fix: default bench line (configs[2] + general_scene block), and where 1-spp / 16-spp frames of configs[3], [2], [1] go by kernel family
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r4_c21_bench.json 2> gpurun_out/r4_c21_bench.err || { tail -5 gpurun_out/r4_c21_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_c21_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("bound", "frac", "frac_alone", "counters_stale")})
print("kernel_ms_per_step", d.get("kernel_ms_per_step"))
print("latency", d.get("latency"))
print("general_scene", d.get("general_scene"))
PY
timeout -k 10 600 python3 tools/latency_breakdown.py 3 2 1 2>&1 | cut -c1-400 | tee gpurun_out/r4_c21_latency.txt
