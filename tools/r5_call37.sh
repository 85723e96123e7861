cd $GRAFT_REPO_ROOT
# the final build's tunables once more (one box, defaults first and last): FH_COOP_T (candidates that trigger a triangle round; default 32), FH_SHADE_WGS (2 | 3 workgroups
# per CU for the shade kernels), FH_TAIL_PATHS (survivors at which the fused tail takes over; default 65536), FH_STREAM_REFILL (24)
run() {  # cfg steps env...
  cfg=$1; steps=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps $steps --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[$cfg] %-28s %9.2f Msamples/s  %9.2f ms  median %9.2f' % ('$*', d['value'], d['ms_per_step'], d['step_ms']['median']))" || exit 1
}
{
for v in FH_X=0 FH_COOP_T=16 FH_COOP_T=24 FH_COOP_T=40 FH_COOP_T=48 FH_SHADE_WGS=2 FH_SHADE_WGS=3 FH_TAIL_PATHS=16384 FH_TAIL_PATHS=262144 FH_STREAM_REFILL=16 FH_STREAM_REFILL=32 FH_X=0; do run 2 8 $v || exit 1; done
for v in FH_X=0 FH_COOP_T=16 FH_COOP_T=24 FH_COOP_T=48 FH_SHADE_WGS=2 FH_TAIL_PATHS=16384 FH_TAIL_PATHS=262144 FH_STREAM_REFILL=16 FH_STREAM_REFILL=32 FH_X=0; do run 3 2 $v || exit 1; done
} > gpurun_out/r5_tunables.log 2>&1; cat gpurun_out/r5_tunables.log
