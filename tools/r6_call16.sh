#!/bin/bash
# round 6, call 16: profile sets of the final build, configs[2] (with the issue-model constants of this fh_trace.h re-measured) and configs[3]
cd $GRAFT_REPO_ROOT
bash tools/r6_profile.sh 2 r06_f issue > gpurun_out/r06_f_profile.log 2>&1 && head -3 gpurun_out/r06_f_profile.log &&
bash tools/r6_profile.sh 3 r06_g3 > gpurun_out/r06_g3_profile.log 2>&1 && head -2 gpurun_out/r06_g3_profile.log
tail -3 gpurun_out/r06_f_profile.log gpurun_out/r06_g3_profile.log | cut -c1-300
