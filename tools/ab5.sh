for ps in 64 128 256; do echo "SPP=256 POOL_SPP=$ps"; SPP=256 POOL_SPP=$ps WORLDS=8,1 python tools/shard_time.py 2>&1 | grep "^world"; done
