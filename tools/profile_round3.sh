#!/bin/bash
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}
# Round-3 profile of one configuration on the GPU box (one gpurun call): the bench line, rocprofv3 kernel traces with the passes in flight and serial,
# and the counter passes (never next to tracing): SQ x2, FETCH_SIZE, WRITE_SIZE + TCC.
# usage: bash tools/profile_round3.sh <tag> <config> [pmc spp]      -> gpurun_out/<tag>_*      then: python tools/collect_profile3.py <tag> <config> <pmc spp>
tag=${1:-r03_x}; cfg=${2:-2}; pspp=${3:-384}; only=${4:-all}   # (4th argument "l1": only the vector-memory passes; "pmc": all counter passes, no bench line and no traces)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --config $cfg ${FH_BENCH_EXTRA:-}"   # (FH_BENCH_EXTRA="--pool-spp 86": pools of the default run's size for counter passes of ONE of its passes)
if [ "$only" = all ]; then
$B > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err; echo "bench rc=$?"
S="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
[ "$cfg" -le 2 ] && S="--steps 6 --warmup 2 --no-cpu-baseline --no-extras"   # (short frames: enough of them that the warm-up frame, whose passes still run with the initial switch depth, does not colour the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- $B $S > $R/gpurun_out/${tag}_stats.log 2>&1
find $R/gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats.csv \;
export FH_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_serial_stats -- $B $S > $R/gpurun_out/${tag}_serial_stats.log 2>&1
find $R/gpurun_out/${tag}_serial_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_serial_kernel_stats.csv \;
find $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_serial_stats -name "*kernel_trace.csv" -delete
fi
# The counter passes render a call of TWO passes of the default run's size, three times (warm-up, timed, the serial step), and a counter file holds per-LAUNCH means over every
# dispatch of the dominant kernel in the process.  For those to be the launches of the default run's passes (round 6, r6-15) the call has to be a multi-pass call like the
# default run's -- a ONE-pass call (what these passes rendered up to r6-14) merges its traversal launches (k_trace_merged_stream), leaves its secondary queues unsorted and picks the
# small calls' flush threshold, and what it left to the dominant kernel were the smallest launches of all -- and the take-over depth of the fused tail is pinned to the one such
# passes settle at (FH_TAIL_DEPTH; probed here: the first pass of a process would otherwise run its cold choice).  Every pass of the process then launches the same set of bounces.
# (--pool-gb 32: with FH_PIPELINE=0 there is ONE pool, and a third of the default run's 96 GB makes it the size of one of the default run's three.)
export FH_PIPELINE=0
wd=$(FH_DEBUG_TAIL=1 $B --steps 1 --warmup 1 --spp $((2 * pspp)) --pool-gb 32 --no-cpu-baseline --no-extras 2>&1 > /dev/null | grep "^\[tail\]" | tail -1 | sed 's/.* wd \([0-9]*\) .*/\1/')
echo "counter passes: --spp $((2 * pspp)) (two passes), take-over depth $wd"; echo ${wd:-0} > $R/gpurun_out/${tag}_wd.txt
run() { name=$1; shift
  FH_TAIL_DEPTH=${wd:-0} timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- $B --steps 1 --warmup 1 --spp $((2 * pspp)) --pool-gb 32 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_$name.log 2>&1; echo "pmc $name rc=$?"
}
if [ "$only" = all ] || [ "$only" = pmc ]; then
run sqa SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD
run sqb SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run tcc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
fi
# the vector-memory side (two counters of a block per pass: more "exceeds the capabilities of the hardware"), with the cycles of the same passes
export FH_PIPELINE=0
run l1a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run l1b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
cd $R
python3 tools/pmc_summary.py "gpurun_out/${tag}_l1?/**/*counter_collection.csv" > gpurun_out/${tag}_l1_summary.txt 2>&1
find gpurun_out/${tag}_l1? -name "*counter_collection.csv" -delete
[ "$only" = all ] || [ "$only" = pmc ] || exit 0
python3 tools/pmc_summary.py "gpurun_out/${tag}_sq?/**/*counter_collection.csv" "gpurun_out/${tag}_fetch/**/*counter_collection.csv" "gpurun_out/${tag}_tcc/**/*counter_collection.csv" > gpurun_out/${tag}_pmc_summary.txt 2>&1
find gpurun_out/${tag}_sq? gpurun_out/${tag}_fetch gpurun_out/${tag}_tcc -name "*counter_collection.csv" -delete
cut -c1-600 gpurun_out/${tag}_bench.json; echo; head -12 gpurun_out/${tag}_serial_kernel_stats.csv | cut -c1-200
