#!/bin/bash
# round 6, call 28: the counter passes again with a clean launch population (profile_round3.sh: FH_MERGE=0, the tail's take-over depth pinned to the steady one); r6-15
# usage: bash tools/r6_call28.sh "<tag> <config> <pspp>" ...
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  set -- $a
  bash tools/profile_round3.sh $1 $2 $3 pmc > gpurun_out/$1_pmc_only.log 2>&1 || { tail -5 gpurun_out/$1_pmc_only.log; exit 1; }
  grep "take-over depth\|rc=" gpurun_out/$1_pmc_only.log | tr '\n' ' '; echo
done
