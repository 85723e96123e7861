cd $GRAFT_REPO_ROOT
bash tools/r5_profile.sh 2 r05_f issue > gpurun_out/r05_f_profile.log 2>&1 && head -2 gpurun_out/r05_f_profile.log &&
bash tools/r5_profile.sh 3 r05_g3 > gpurun_out/r05_g3_profile.log 2>&1 && head -1 gpurun_out/r05_g3_profile.log &&
bash tools/r5_profile.sh 1 r05_f1 > gpurun_out/r05_f1_profile.log 2>&1 && head -1 gpurun_out/r05_f1_profile.log &&
bash tools/r5_profile.sh 4 r05_f4 > gpurun_out/r05_f4_profile.log 2>&1 && head -1 gpurun_out/r05_f4_profile.log
