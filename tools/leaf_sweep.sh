run() { echo "$@"; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['kernel_ms_per_step'], d['roofline']['per_ray'], d['bvh'])"; }
run FH_LEAF8=1
run FH_LEAF8=2
run FH_LEAF8=3
