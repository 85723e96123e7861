cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python tools/shard_time.py > gpurun_out/r5_c9_shards.jsonl 2> gpurun_out/r5_c9_shards.err; tail -2 gpurun_out/r5_c9_shards.err
