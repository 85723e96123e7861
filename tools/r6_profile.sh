#!/bin/bash
# round 5: the profile set of one configuration in one gpurun call.  The counter passes (which run with FH_PIPELINE=0: one pool, every kernel alone) render ONE pass of the size
# the configuration's default run submits (spp per step / passes per step of its own line), so that bench.py may use the counter file for the default run (usable_counters:
# same pass size or not at all).
# usage: bash tools/r5_profile.sh <config> <tag> [issue]     ("issue": first re-measure the issue-model constants of this fh_trace.h -> profiles/r06_issue_peak.json)
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}
cd $GRAFT_REPO_ROOT
cfg=$1; tag=$2
if [ "${3:-}" = issue ]; then
  h=$(sha256sum fredholm_amd/csrc/fh_trace.h | cut -c1-16)
  timeout -k 10 200 tools/micro/issue_peak.bin --json gpurun_out/r06_issue_peak.json $h 60 > gpurun_out/r06_issue_peak.txt 2>&1 || { tail -5 gpurun_out/r06_issue_peak.txt; exit 1; }
  cp gpurun_out/r06_issue_peak.json profiles/r06_issue_peak.json; cat gpurun_out/r06_issue_peak.json
fi
pspp=$(python3 bench.py --config $cfg --no-cpu-baseline --no-extras --steps 1 --warmup 0 2>/dev/null | python3 -c "
import json,sys
c=json.loads(sys.stdin.read().strip().splitlines()[-1])['config']
print(max(int(round(c['spp_per_step']/max(c['passes_per_step'],1))),1))") || exit 1
echo "counter passes: --spp $pspp"; echo $pspp > gpurun_out/${tag}_pspp.txt
bash tools/profile_round3.sh $tag $cfg $pspp
