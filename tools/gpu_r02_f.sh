#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_f_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_f_pytest.log | cut -c1-300
for lib in default shade2; do for c in 2 1; do
  if [ $lib = default ]; then unset FH_LIB; else export FH_LIB=$GRAFT_REPO_ROOT/fredholm_amd/libfredholm_hip_$lib.so; fi
  python3 bench.py --config $c --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r02_f_${lib}_config$c.json 2> gpurun_out/r02_f_${lib}_config$c.err; echo "$lib config $c rc=$?"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r02_f_${lib}_config$c.json"))
print("$lib", $c, d["value"], d["step_ms"]["median"], d["kernel_ms_per_step_alone"])
PY
done; done
