#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 8192 16384 32768 65536 131072 262144 1048576; do
  FH_TAIL_PATHS=$t python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_tail_$t.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/sweep_tail_$t.json")); k=d["kernel_ms_per_step_alone"]
print("tail paths $t: %.0f Msamples/s  median %.1f ms  alone: closest %.1f secondary %.1f shade %.1f tail %.1f sort %.1f total %.1f" % (d["value"], d["step_ms"]["median"], k["trace_closest"], k["trace_secondary"], k["shade"], k["tail"], k["route_and_sort"], k["render_total"]), flush=True)
PY
done
for c in 32 128 256; do
  FH_STREAM_CHUNK=$c python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_chunk_$c.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("gpurun_out/sweep_chunk_$c.json")); k=d["kernel_ms_per_step_alone"]
print("chunk $c: %.0f Msamples/s  closest %.1f secondary %.1f" % (d["value"], k["trace_closest"], k["trace_secondary"]), flush=True)
PY
done
