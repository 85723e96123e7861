#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_e_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_e_pytest.log | cut -c1-400
for c in 3 1 2; do
  python3 bench.py --config $c --no-cpu-baseline $( [ $c = 3 ] && echo "--spp 512 --steps 2" ) > gpurun_out/r02_e_bench_config$c.json 2> gpurun_out/r02_e_bench_config$c.err; echo "config $c rc=$?"; tail -c 300 gpurun_out/r02_e_bench_config$c.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r02_e_bench_config$c.json"))
print($c, d["value"], d["step_ms"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["avg_launch_ms_alone"], d["roofline"]["measured_hbm_gbs"], d["kernel_ms_per_step_alone"], d["rates"])
PY
done
