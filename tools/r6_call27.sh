#!/bin/bash
# round 6, call 27: the tail's depth picked from survival SHARES (call-size changes), idle tail blocks leaving before the staging, ray records without indeterminate fields
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_27_tail_depth_shares.log; : > $out
for v in "" "FH_TAIL_DEPTH=1"; do
  env $v timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_27_tests.log 2>&1; rc=$?
  echo "parity ${v:-default}: rc $rc $(tail -1 gpurun_out/r6_27_tests.log)" >> $out
  [ $rc -eq 0 ] || { cat $out; grep -n "FAILED\|Error" gpurun_out/r6_27_tests.log | head -5; exit 1; }
done
timeout -k 10 300 python tools/call_size_change.py 3 2 >> $out 2>&1 &&
timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1 &&
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r6_27_bench.json 2>> $out
python - >> $out <<'PY'
import json
d = json.loads(open("gpurun_out/r6_27_bench.json").read().strip().splitlines()[-1])
print("configs[2]", d["value"], "general_scene", (d.get("general_scene") or {}).get("value"))
PY
cat $out
