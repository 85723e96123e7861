run() { echo "$@"; env $1 timeout 300 python bench.py --no-cpu-baseline --steps 8 $2 $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"; }
run FH_PIPELINE=0 --pool-spp 64
run FH_PIPELINE=1 --pool-spp 64
run FH_PIPELINE=1 --pool-spp 32
run FH_PIPELINE=1 --pool-spp 16
run FH_PIPELINE=0 --pool-spp 16
run FH_PIPELINE=1 --pool-spp 8
