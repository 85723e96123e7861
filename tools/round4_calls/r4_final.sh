#!/bin/bash
# round 4, final call: the GPU suite as the driver runs it, smoke(), and the default bench line with the round's counter files in place
cd $GRAFT_REPO_ROOT
set -o pipefail
PYTHONFAULTHANDLER=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04_final_gpu_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r04_final_gpu_tests.log; [ $rc -eq 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > gpurun_out/r04_h_bench.json 2> gpurun_out/r04_h_bench.err || { tail -5 gpurun_out/r04_h_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_h_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", r["frac"], r.get("frac_alone"), "stale", r.get("counters_stale"), "issue stale", r["issue_model"].get("stale"), "traffic", r.get("traffic"), "frac_hbm_measured", r.get("frac_hbm_measured"))
print("kernel_info", r.get("kernel_info"))
print("latency", d["latency"]["spp1"], d["latency"]["spp16"])
g = d["general_scene"]; gr = g["roofline"]
print("general", g["msamples_per_s"], gr["frac"], gr.get("frac_alone"), gr.get("counters_stale"), gr.get("kernel_info"), g["latency"]["spp1"], g["latency"]["spp16"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity"]["bit_identical_pixels"])
PY
