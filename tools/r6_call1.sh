#!/bin/bash
# round 6, call 1: the GPU suite on the build with (a) fh_kat_bsdf_ior + the third-table replay, (b) zero-contribution secondary rays dropped, (c) one-pass calls no longer
# counted as probes of the ray-start decision; then the BEFORE timelines of the reference's call pattern on configs[3] (1 and 16 samples per call) and configs[1]'s throughput.
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_1_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r6_1_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spp in 1 16; do
  rm -rf gpurun_out/tl_$spp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$spp -o tl -- python3 tools/call_timeline.py run 3 $spp 40 > gpurun_out/r6_1_tl_run_$spp.log 2>&1 || { tail -5 gpurun_out/r6_1_tl_run_$spp.log; exit 1; }
  tail -1 gpurun_out/r6_1_tl_run_$spp.log
  f=$(find gpurun_out/tl_$spp -name "*kernel_trace.csv" | head -1)
  python3 tools/call_timeline.py reduce $f $spp 40 > gpurun_out/r6_before_timeline_config3_${spp}spp.txt && tail -4 gpurun_out/r6_before_timeline_config3_${spp}spp.txt
  rm -rf gpurun_out/tl_$spp
done
timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 > gpurun_out/r6_1_latency.log 2>&1; cat gpurun_out/r6_1_latency.log
timeout -k 10 300 python bench.py --config 1 --no-cpu-baseline --no-extras > gpurun_out/r6_1_bench_c1.json 2> gpurun_out/r6_1_bench_c1.err; python3 -c "
import json; j=json.loads(open('gpurun_out/r6_1_bench_c1.json').read().strip().splitlines()[-1]); print('configs[1]', j['value'], j['ms_per_step'], j.get('rates'))"
