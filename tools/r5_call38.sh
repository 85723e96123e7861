cd $GRAFT_REPO_ROOT
# second pass over the tunables that moved in r5_call37: combinations, and the other configurations
run() {  # cfg steps env...
  cfg=$1; steps=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps $steps --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[$cfg] %-52s %9.2f Msamples/s  %9.2f ms  median %9.2f' % ('$*', d['value'], d['ms_per_step'], d['step_ms']['median']))" || exit 1
}
{
run 2 8 FH_X=0 && run 2 8 FH_SHADE_WGS=3 FH_COOP_T=48 && run 2 8 FH_SHADE_WGS=3 FH_COOP_T=48 FH_TAIL_PATHS=16384 && run 2 8 FH_SHADE_WGS=3 FH_COOP_T=64 && run 2 8 FH_X=0 &&
run 3 2 FH_X=0 && run 3 2 FH_COOP_T=48 && run 3 2 FH_COOP_T=64 && run 3 2 FH_COOP_T=48 FH_TAIL_PATHS=16384 && run 3 2 FH_COOP_T=48 FH_TAIL_PATHS=8192 &&
run 4 1 FH_X=0 && run 4 1 FH_SHADE_WGS=3 FH_COOP_T=48 && run 4 1 FH_SHADE_WGS=3 FH_COOP_T=48 FH_TAIL_PATHS=16384 &&
run 1 4 FH_X=0 && run 1 4 FH_COOP_T=48 && run 1 4 FH_SHADE_WGS=3
} > gpurun_out/r5_tunables2.log 2>&1; rc=$?; cat gpurun_out/r5_tunables2.log; exit $rc
