#!/bin/bash
# round 6, call 17: profile sets of the final build, configs[1] and configs[4]
cd $GRAFT_REPO_ROOT
bash tools/r6_profile.sh 1 r06_f1 > gpurun_out/r06_f1_profile.log 2>&1 && head -2 gpurun_out/r06_f1_profile.log &&
bash tools/r6_profile.sh 4 r06_f4 > gpurun_out/r06_f4_profile.log 2>&1 && head -2 gpurun_out/r06_f4_profile.log
cat gpurun_out/r06_f1_pspp.txt gpurun_out/r06_f4_pspp.txt
