#!/bin/bash
# kernel-trace and HIP-event timing of the same command with the two-passes-in-flight pipelining switched off (kernels run alone)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
export FH_PIPELINE=0
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_serial_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_serial_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_serial_stats.log 2>&1
find $R/gpurun_out/${tag}_serial_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_serial_kernel_stats.csv \;
find $R/gpurun_out/${tag}_serial_stats -name "*kernel_trace.csv" -delete
cut -c1-500 $R/gpurun_out/${tag}_serial_bench.json; grep "k_trace_secondary_stream<false" $R/gpurun_out/${tag}_serial_kernel_stats.csv | cut -c1-60,200-300
