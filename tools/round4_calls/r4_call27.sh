#!/bin/bash
# round 4, call 27: the sky-pixel kernel on a stream of the lowest priority (FH_SKY_PRIO=1), alone and with a bounded grid; configs[2] and [4]
cd $GRAFT_REPO_ROOT
ab() { cfg=$1; extra=$2; shift 2; for v in "$@"; do tag=$(echo "$v" | tr ' =' '__'); env $v timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $extra > gpurun_out/env2_${tag}_$cfg.json 2> gpurun_out/env2_${tag}_$cfg.err || { echo "$v FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/env2_${tag}_$cfg.json')); a=d['kernel_ms_per_step_alone']
print('$v config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; alone generate', a['generate'], 'closest', a['trace_closest'], 'secondary', a['trace_secondary'])"; done; }
echo "== configs[2]"; ab 2 "--steps 6 --warmup 2" "FH_X=0" "FH_SKY_PRIO=1" "FH_SKY_PRIO=1 FH_SKY_BLOCKS=2" "FH_SKY_BLOCKS=1" "FH_X=0" "FH_SKY_PRIO=1"
echo "== configs[4], 1024 spp"; ab 4 "--spp 1024 --steps 2 --warmup 1" "FH_X=0" "FH_SKY_PRIO=1" "FH_SKY_BLOCKS=1"
