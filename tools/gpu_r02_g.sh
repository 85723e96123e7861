#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_g_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02_g_pytest.log | cut -c1-300
for c in 2 1 3; do
  python3 bench.py --config $c --no-cpu-baseline --steps 4 --warmup 1 $( [ $c = 3 ] && echo "--spp 512 --steps 2" ) > gpurun_out/r02_g_config$c.json 2> gpurun_out/r02_g_config$c.err; echo "config $c rc=$?"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r02_g_config$c.json"))
print($c, d["value"], d["step_ms"]["median"], d["kernel_ms_per_step_alone"])
PY
done
