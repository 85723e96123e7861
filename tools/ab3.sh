for ps in 64 32 16; do for pl in 0 1; do echo "POOL_SPP=$ps FH_PIPELINE=$pl"; POOL_SPP=$ps FH_PIPELINE=$pl WORLDS=8,4 python tools/shard_time.py 2>&1 | grep "^world"; done; done
