cd $GRAFT_REPO_ROOT
# A/B of two builds of the library on one box: HEAD (spill column behind a vector asm barrier) against the working tree (column from scalars, FH_BOTTOM_UP_ALPHA=0)
: > gpurun_out/r5_ab35.log
for cfg in 3 2; do
  for lib in head new head new; do
    echo "== configs[$cfg] $lib" >> gpurun_out/r5_ab35.log
    FH_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_$lib.so timeout -k 10 300 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps $([ $cfg = 3 ] && echo 2 || echo 8) --warmup 1 2> gpurun_out/r5_ab35_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d.get('kernel_ms_per_step_alone') or {}; print(d['value'], d['ms_per_step'], d['step_ms'], {k: a.get(k) for k in ('trace_closest','trace_secondary','shade')})" >> gpurun_out/r5_ab35.log || { tail -5 gpurun_out/r5_ab35_err.log; exit 1; }
  done
done
cat gpurun_out/r5_ab35.log
