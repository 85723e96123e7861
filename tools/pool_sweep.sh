#!/bin/bash
# throughput against the memory the path pools may take: tools/pool_sweep.sh "<GB ...>" "<configs>"  -> gpurun_out/pool_sweep_config<N>.jsonl (one bench line per budget)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in $2; do
  : > gpurun_out/pool_sweep_config$cfg.jsonl
  for gb in $1; do
    extra="--pool-gb $gb"; [ "$gb" = default ] && extra=""
    timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $extra $3 2> gpurun_out/pool_sweep.err | tail -1 > gpurun_out/pool_sweep_line.json || { echo "config $cfg $gb GB FAILED"; tail -3 gpurun_out/pool_sweep.err; continue; }
    python3 - "$gb" $cfg <<'PY'
import json, sys
d = json.load(open("gpurun_out/pool_sweep_line.json"))
row = {"config": int(sys.argv[2]), "budget_gb": sys.argv[1], "pools_gb": d["config"]["path_pools"]["gb"], "pool_paths": d["config"]["path_pools"]["paths"], "passes_per_step": d["config"]["passes_per_step"],
       "msamples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "step_ms": d["step_ms"]}
open("gpurun_out/pool_sweep_config%d.jsonl" % row["config"], "a").write(json.dumps(row) + "\n")
print(row, flush=True)
PY
  done
done
