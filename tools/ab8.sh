run() { echo "$@"; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 3 --spp 256 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; }
run FH_PIPELINE=0
run FH_PIPELINE=0 FH_LIB=$PWD/fredholm_amd/libfredholm_hip_clds.so
run FH_PIPELINE=0
run FH_PIPELINE=0 FH_LIB=$PWD/fredholm_amd/libfredholm_hip_clds.so
