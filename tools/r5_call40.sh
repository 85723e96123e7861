cd $GRAFT_REPO_ROOT
# the small calls of the soup with the per-call flush threshold (r5-13), then the profile sets of configs 2 / 3 / 1 on this build (configs[4] keeps r5-12's files)
timeout -k 10 120 python tools/latency_small_calls.py 2 2>/dev/null | cut -c1-200 > gpurun_out/r5_latency_after.log || exit 1
cat gpurun_out/r5_latency_after.log
bash tools/r5_profile.sh 2 r05_f issue > gpurun_out/r05_f_profile.log 2>&1 && head -2 gpurun_out/r05_f_profile.log &&
bash tools/r5_profile.sh 3 r05_g3 > gpurun_out/r05_g3_profile.log 2>&1 && head -1 gpurun_out/r05_g3_profile.log &&
bash tools/r5_profile.sh 1 r05_f1 > gpurun_out/r05_f1_profile.log 2>&1 && head -1 gpurun_out/r05_f1_profile.log
