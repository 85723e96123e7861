#!/bin/bash
# round 6, call 19: rocprofv3's kernel trace of the bench command itself, reduced to the launches of the timed region (tools/timed_region_avg.py): does the profiler's average
# duration of the dominant kernel agree with the HIP-event figure of the line?
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr -o tr -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras --no-general-scene > gpurun_out/r06_trace_line.json 2> gpurun_out/r6_19.err || { tail -5 gpurun_out/r6_19.err; exit 1; }
f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timed_region_avg.py $f gpurun_out/r06_trace_line.json | tee gpurun_out/r06_timed_region_avg.json
find gpurun_out/tr -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_trace_kernel_stats.csv \;
rm -rf gpurun_out/tr
