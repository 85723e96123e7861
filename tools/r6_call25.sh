#!/bin/bash
# round 6, call 25: the streamed tail as its own instantiation (every cooperative trace of the tail through TailStream; the rounds form compiled separately, FH_TAIL_STREAM=0)
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_25_tail_stream.log; : > $out
for v in "" "FH_TAIL_PATHS=1024" "FH_TAIL_DEPTH=1"; do
  env $v timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_25_tests.log 2>&1; rc=$?
  echo "parity ${v:-default}: rc $rc $(tail -1 gpurun_out/r6_25_tests.log)" >> $out
  [ $rc -eq 0 ] || { cat $out; grep -n "FAILED\|Error" gpurun_out/r6_25_tests.log | head -5; exit 1; }
done
for v in "" FH_TAIL_STREAM=0 "" FH_TAIL_STREAM=0; do
  echo "== ${v:-streamed (default)}" >> $out
  env $v timeout -k 10 400 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1
done
cat $out
