#!/bin/bash
# SQ counter passes (VALU issue, LDS, waits) for the final-build kernels; one group per pass, no tracing next to --pmc.
# usage (on the GPU box): bash tools/profile_sq.sh <tag> [extra bench args]   -> gpurun_out/<tag>_sq_summary.txt
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- python3 $R/bench.py --steps 1 --warmup 1 --spp 256 --no-cpu-baseline $EXTRA > $R/gpurun_out/${tag}_$name.log 2>&1
}
EXTRA="$*"
run sqa SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD &&
run sqb SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
cd $R && python3 tools/pmc_summary.py "gpurun_out/${tag}_sq?/**/*counter_collection.csv" > gpurun_out/${tag}_sq_summary.txt 2>&1
find gpurun_out/${tag}_sq? -name "*counter_collection.csv" -delete
head -40 gpurun_out/${tag}_sq_summary.txt
