#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import fredholm_amd as F
r=F.Renderer(0)
for nb in (1<<28, 1<<30, 1<<31):
    print("bandwidth", nb>>20, "MiB buffers:", [round(x,1) for x in r.measure_bandwidth(nb, 6)])
r.close()
PY
for lib in default gen4 gen8; do
  if [ $lib = default ]; then unset FH_LIB; else export FH_LIB=$GRAFT_REPO_ROOT/fredholm_amd/libfredholm_hip_$lib.so; fi
  python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r02_l_$lib.json 2> gpurun_out/r02_l_$lib.err; echo "$lib rc=$?"
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r02_l_$lib.json"))
print("$lib", d["value"], d["step_ms"]["median"], d["kernel_ms_per_step_alone"]["generate"], d["roofline"].get("valu_issue_frac"), d["roofline"].get("hbm_traffic_frac"), d["roofline"]["measured_hbm_gbs"])
PY
done
unset FH_LIB
python3 bench.py --config 4 --no-cpu-baseline --spp 512 --steps 2 > gpurun_out/r02_l_config4.json 2> gpurun_out/r02_l_config4.err; python3 -c "
import json; d=json.load(open('gpurun_out/r02_l_config4.json')); print(d['value'], d['post_chain'])"
