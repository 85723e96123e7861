#!/bin/bash
# round 4, call 24: (a) configs[3] after the tiling experiment was taken out again; (b) per-kernel times of configs[2] with the passes serial (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT
bash tools/gpu_env_ab.sh "FH_X=0" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
FH_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_c24_serial -- python3 $R/bench.py --config 2 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r4_c24_serial.log 2>&1
find $R/gpurun_out/r4_c24_serial -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r4_c24_serial_kernel_stats.csv \;
find $R/gpurun_out/r4_c24_serial -name "*kernel_trace.csv" -delete
head -16 $R/gpurun_out/r4_c24_serial_kernel_stats.csv | cut -c1-180
