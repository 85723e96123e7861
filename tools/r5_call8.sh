cd $GRAFT_REPO_ROOT
bash tools/pool_sweep.sh "default 128 96 64 48 32 16 8 default" "2" "--steps 8" > gpurun_out/r5_c8_sweep2.log 2>&1
bash tools/pool_sweep.sh "default 128 64 32 16 8" "3" "--spp 1024 --steps 2" > gpurun_out/r5_c8_sweep3.log 2>&1
