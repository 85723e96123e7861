#!/bin/bash
# round 6, call 10: LDS bank conflicts of the cooperative triangle test.  The tester's copy of a lane's ray as four 8-byte columns (variant "cols",
# -DFH_COOP_RAY_COLUMNS=1: ds_read_b64, 32-lane groups, owners o and o + 32 collide: 2-way at most) against two 16-byte rows (ds_read_b128, 16-lane groups, four rows per
# bank class).  Time: same-box A/B on configs[3] (540 spp) and [2]; counters: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel, one --pmc pass per variant and configuration.
: ${GRAFT_REPO_ROOT:?run on the GPU box}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_10_lds_columns.log; : > $out
FH_LIB=$PWD/fredholm_amd/libfredholm_hip_cols.so timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1 || echo "variant cols: smoke FAILED" >> $out
for cfg in 3 2; do
  for v in base cols base cols; do
    lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
    spp=""; [ $cfg = 3 ] && spp="--spp 540"
    FH_LIB=$PWD/$lib timeout -k 10 300 python bench.py --config $cfg $spp --no-cpu-baseline --no-extras --no-general-scene 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=j['kernel_ms_per_step_alone']; print('configs[$cfg] $v:', j['value'], 'Msamples/s', j['ms_per_step'], 'ms; alone closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'shade', a['shade'], 'total', a['render_total'])" >> $out
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 3 2; do
  spp=45; [ $cfg = 2 ] && spp=128
  for v in base cols; do
    lib=fredholm_amd/libfredholm_hip.so; [ "$v" != base ] && lib=fredholm_amd/libfredholm_hip_$v.so
    rm -rf gpurun_out/pmc_lds_${v}_$cfg
    FH_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds_${v}_$cfg -- python3 bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --no-cpu-baseline --no-extras --no-general-scene > gpurun_out/r6_10_pmc_${v}_$cfg.log 2>&1 || { echo "pmc $v $cfg failed"; tail -3 gpurun_out/r6_10_pmc_${v}_$cfg.log; }
    echo "---- configs[$cfg] $v: LDS counters per kernel (mean per dispatch)" >> $out
    python3 tools/pmc_summary.py "gpurun_out/pmc_lds_${v}_$cfg/**/*counter_collection.csv" | python3 -c "
import sys,re
blk=None; d={}
for ln in sys.stdin:
    if ln.startswith('k_'): blk=ln.split('  dispatches=')[0]; d[blk]={}
    else:
        m=re.match(r'\s+(\w+)\s+mean\s+([\d.]+)',ln)
        if m and blk: d[blk][m.group(1)]=float(m.group(2))
for k,v in d.items():
    if k.startswith('k_trace') and not k.startswith(('k_trace_closest_stream<true','k_trace_secondary_stream<true')) and v.get('SQ_LDS_IDX_ACTIVE'):
        print('  %-48s conflict / active = %.3f   (LDS-active cycles / busy cycles %.3f)' % (k, v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE'], v['SQ_LDS_IDX_ACTIVE']/max(v.get('SQ_BUSY_CYCLES',0),1)))" >> $out
    rm -rf gpurun_out/pmc_lds_${v}_$cfg
  done
done
cat $out
