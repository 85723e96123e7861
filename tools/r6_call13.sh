#!/bin/bash
# round 6, call 13: from how many camera paths on should a call that fits one pass be cut into three overlapping ones?  (default 3 << 24 = 50 M: rank 0's shard of configs[2]
# -- 38 M paths after the sky split -- runs as ONE pass and takes 38.6 ms where the whole frame / 8 is 33.)  Shards and small whole-frame calls under FH_THREE_PASS_MIN.
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_13_three_pass_min.log; : > $out
for v in 50331648 25165824 12582912 6291456 3145728 50331648; do
  echo "== FH_THREE_PASS_MIN=$v" >> $out
  FH_THREE_PASS_MIN=$v STEPS=5 timeout -k 10 300 python tools/shard_pass_probe.py 2>/dev/null | grep "bench default" >> $out
  FH_THREE_PASS_MIN=$v timeout -k 10 300 python tools/latency_small_calls.py 3 2 1 >> $out 2>&1
done
cat $out
