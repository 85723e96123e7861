#!/bin/bash
# Round profile on the GPU box: kernel trace + the two HBM counter passes (separate runs, as the MI355X guide prescribes).
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,tcc}/, <tag>_pmc_summary.txt, <tag>_bench.json
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${tag}_tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_tcc.log 2>&1
cd $R
python3 tools/pmc_summary.py "gpurun_out/${tag}_fetch/**/*counter_collection.csv" "gpurun_out/${tag}_tcc/**/*counter_collection.csv" > gpurun_out/${tag}_pmc_summary.txt
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
# the raw per-dispatch traces are large; keep the summaries only
find gpurun_out/${tag}_stats -name "*kernel_trace.csv" -delete
tail -c 600 gpurun_out/${tag}_bench.json; head -12 gpurun_out/${tag}_kernel_stats.csv
