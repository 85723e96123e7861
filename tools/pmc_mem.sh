#!/bin/bash
# PMC passes for the memory pipeline (TA / TCP / TD) of the traversal kernels; one small counter group per pass.
tag=${1:-mem}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_$name.log 2>&1
}
run p1 TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run p2 TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run p3 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
run p4 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
run p5 TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_BUSY_CU_CYCLES
run p6 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
cd $R && python3 tools/pmc_summary.py "gpurun_out/${tag}_p*/**/*counter_collection.csv" > gpurun_out/${tag}_summary.txt 2>&1
tail -5 gpurun_out/${tag}_p1.log
