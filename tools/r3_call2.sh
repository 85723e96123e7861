#!/bin/bash
# round 3, call 2: microbenchmark v2 (co-issue, SDWA), refill sweep on the 64-byte node layout, small-launch latency baseline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 tools/micro/issue_peak.bin 60 > gpurun_out/r03_issue_peak.txt 2>&1; echo "issue_peak rc $?"; head -45 gpurun_out/r03_issue_peak.txt
for r in 8 16 24 32 40; do
  FH_STREAM_REFILL=$r timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/refill_$r.json 2> gpurun_out/refill_$r.err || { echo "refill $r FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/refill_$r.json')); a=d['kernel_ms_per_step_alone']
print('refill $r:', d['value'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'])"
done
timeout -k 10 300 python3 tools/frame_latency.py > gpurun_out/latency_base.txt 2>&1; cat gpurun_out/latency_base.txt
