#!/bin/bash
# round 4, call 6: the shade side of a pass on its own stream (normal / high priority), with and without a CU mask on the pass streams; off / on on the same box
cd $GRAFT_REPO_ROOT
run() { # name, config, extra args, env...
  name=$1; cfg=$2; extra=$3; shift 3
  env "$@" timeout -k 10 400 python3 bench.py --config $cfg --no-cpu-baseline --no-extras $extra > gpurun_out/ss_${name}_$cfg.json 2> gpurun_out/ss_${name}_$cfg.err || { echo "$name config $cfg FAILED"; tail -3 gpurun_out/ss_${name}_$cfg.err; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/ss_${name}_$cfg.json'))
a=d['kernel_ms_per_step_alone']; t=d['kernel_ms_per_step']
print('$name config $cfg:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shade span', t['shade'], 'alone', a['shade'], '; closest span', t['trace_closest'], 'secondary span', t['trace_secondary'])"
}
for cfg in 1 2; do
  S="--steps 6 --warmup 2"; [ $cfg = 1 ] && S="--steps 4 --warmup 1"
  run off $cfg "$S" FH_X=0
  run stream $cfg "$S" FH_SHADE_STREAM=1
  run prio $cfg "$S" FH_SHADE_STREAM=2
  run prio_cu28 $cfg "$S" FH_SHADE_STREAM=2 FH_TRACE_CUS=28
  run prio_cu24 $cfg "$S" FH_SHADE_STREAM=2 FH_TRACE_CUS=24
  run off $cfg "$S" FH_X=0
done
run off 3 "--spp 512 --steps 2 --warmup 1" FH_X=0
run prio 3 "--spp 512 --steps 2 --warmup 1" FH_SHADE_STREAM=2
run prio_cu28 3 "--spp 512 --steps 2 --warmup 1" FH_SHADE_STREAM=2 FH_TRACE_CUS=28
