#!/bin/bash
# round 6, call 12: the emulated 8-shard split again on this round's build, now with rank 0's share of a presented frame (pack, one-launch un-permutation, post chain) timed
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CONFIGS="2,3" SPPS="1024,16" TILES="32" STEPS=4 timeout -k 10 900 python tools/shard_time.py > gpurun_out/r06_shard_times.jsonl 2> gpurun_out/r6_12_shard.err || { tail -5 gpurun_out/r6_12_shard.err; exit 1; }
CONFIGS="4" SPPS="256" TILES="32" STEPS=3 timeout -k 10 600 python tools/shard_time.py >> gpurun_out/r06_shard_times.jsonl 2>> gpurun_out/r6_12_shard.err || { tail -5 gpurun_out/r6_12_shard.err; exit 1; }
python3 -c "
import json
for ln in open('gpurun_out/r06_shard_times.jsonl'):
    j=json.loads(ln); print(j['config'], j['spp'], 'whole', j['whole_ms'], 'max shard', j['max_ms'], 'render x', j['render_speedup_whole_over_max'], 'sink', j['rank0_sink'], 'step x', j['step_speedup'])"
