#!/bin/bash
# round 2, call d: full-size config tests + bench lines for configs 1..4
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02_d_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_d_pytest.log
for c in 1 3 4 2; do
  python3 bench.py --config $c > gpurun_out/r02_d_bench_config$c.json 2> gpurun_out/r02_d_bench_config$c.err; echo "config $c rc=$?"; tail -c 400 gpurun_out/r02_d_bench_config$c.err
  cut -c1-1800 gpurun_out/r02_d_bench_config$c.json
done
