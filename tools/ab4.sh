run() { echo "$@"; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"; }
for c in 64 128 256 512; do run FH_STREAM_CHUNK=$c FH_PIPELINE=0; done
run FH_STREAM_CHUNK=128 FH_PIPELINE=1
for c in 64 128 256; do echo "chunk $c"; FH_STREAM_CHUNK=$c FH_PIPELINE=0 WORLDS=8 python tools/shard_time.py 2>&1 | grep -A1 "^world"; done
