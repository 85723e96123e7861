#!/bin/bash
# PMC collection for the traversal kernels, one counter group per pass (rocprofv3 --pmc must not be combined with tracing).
# usage (on the GPU box): bash tools/pmc_passes.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run tcc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']; tag=os.environ.get('TAG','pmc')
PY
