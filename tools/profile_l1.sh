#!/bin/bash
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}
# vector-memory side of the traversal kernels (one gpurun call): texture-address / L1 / TLB counters, each group in its own pass.
# usage: bash tools/profile_l1.sh <tag> <config> [pmc spp]      -> gpurun_out/<tag>_l1_summary.txt
tag=${1:-r03_l1}; cfg=${2:-2}; pspp=${3:-384}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --config $cfg"
run() { name=$1; shift
  timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- $B --steps 1 --warmup 1 --spp $pspp --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_$name.log 2>&1; echo "pmc $name rc=$?"
}
# (two counters of a block per pass: more "exceeds the capabilities of the hardware")
run a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run c TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run d TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run e TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
cd $R
python3 tools/pmc_summary.py "gpurun_out/${tag}_[a-e]/**/*counter_collection.csv" > gpurun_out/${tag}_l1_summary.txt 2>&1
find gpurun_out/${tag}_[a-e] -name "*counter_collection.csv" -delete
grep -A14 "^k_trace_secondary_stream<false\|^k_trace_closest_stream<false" gpurun_out/${tag}_l1_summary.txt | cut -c1-120
