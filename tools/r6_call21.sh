#!/bin/bash
# round 6, call 21: the N > 1 step with FOUR ranks (gloo, sharing this one GPU; pools bounded), started by bench.py itself, --check-frame: the gathered frame of a 4-way tile
# split bit-identical to the unsharded render.  (Two ranks are in the suite; eight would exceed the box's limit of six GPU processes.)
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in 2 3; do
  FH_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --config $cfg --steps 2 --warmup 1 --spp 8 --pool-gb 4 --check-frame --no-cpu-baseline > gpurun_out/r06_four_ranks_config$cfg.json 2> gpurun_out/r06_four_ranks_config$cfg.err; echo "configs[$cfg] rc $?"
  grep "check-frame" gpurun_out/r06_four_ranks_config$cfg.err
  python3 -c "
import json; j=json.loads(open('gpurun_out/r06_four_ranks_config$cfg.json').read().strip().splitlines()[-1]); print(j['n_gpus'], j['value'], j['ms_per_step'], j['config']['parallelism'], j['config']['gather'][:60])"
done
