#!/bin/bash
# round 2, call b: parity suite on the new LDS-only stack, VALU issue rates, workgroups-per-CU sweep
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_b_pytest.log
./tools/micro/valu_rate.bin > gpurun_out/r02_b_valu_rate.txt 2>&1; cat gpurun_out/r02_b_valu_rate.txt
for w in 6 5 4 3; do
  FH_STREAM_WGS=$w python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_b_bench_wgs$w.json 2> gpurun_out/r02_b_bench_wgs$w.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r02_b_bench_wgs$w.json"))
print("wgs $w", d["value"], d["kernel_ms_per_step"], d["bvh"], d["roofline"]["avg_launch_ms"])
PY
done
