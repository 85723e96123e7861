run() { echo "$@"; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['kernel_ms_per_step'])"; }
run FH_X=1
for G in 1024 1536 2560 3840; do run FH_STREAM_GRID=$G; done
for R in 8 24 32; do run FH_STREAM_REFILL=$R; done
for T in 2 8 16; do run FH_COOP_T=$T; done
