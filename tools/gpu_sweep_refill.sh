#!/bin/bash
# sweep of the streaming traversal's refill threshold and cooperative flush threshold (results do not depend on either)
cd $GRAFT_REPO_ROOT
for r in 8 16 24 32 40; do for t in 24 32 48; do
  FH_STREAM_REFILL=$r FH_COOP_T=$t python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_r${r}_t${t}.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/sweep_r${r}_t${t}.json"))
k=d["kernel_ms_per_step"]
print("refill $r flush $t: %.0f Msamples/s closest %.1f secondary %.1f tail %.1f" % (d["value"], k["trace_closest"], k["trace_secondary"], k["tail"]), flush=True)
PY
done; done
