cd $GRAFT_REPO_ROOT
# streaming parameters re-swept with rays that start at their face (configs[2], 1024-spp frames): refill threshold, flush threshold, queue chunk of the secondary launch
for v in "FH_AB_OFF=1" "FH_STREAM_REFILL=16" "FH_STREAM_REFILL=32" "FH_COOP_T=24" "FH_COOP_T=40" "FH_STREAM_CHUNK=96" "FH_STREAM_CHUNK=128" "FH_AB_OFF=1"; do
  env $v timeout -k 10 400 python3 bench.py --config 2 --no-cpu-baseline --no-extras --steps 8 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$v FAILED"; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/ab.json')); a=d['kernel_ms_per_step_alone']
print('$v:', d['value'], 'Msamples/s; alone closest', a['trace_closest'], 'secondary', a['trace_secondary'], 'shade', a['shade'])"
done
