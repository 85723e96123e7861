cd $GRAFT_REPO_ROOT
bash tools/pool_sweep.sh "72 96 72 96" "2" "--steps 8" > gpurun_out/r5_c12.log 2>&1
bash tools/pool_sweep.sh "72 96 128" "3" "--spp 512 --steps 2" >> gpurun_out/r5_c12.log 2>&1
bash tools/pool_sweep.sh "72 96" "1" "" >> gpurun_out/r5_c12.log 2>&1
bash tools/pool_sweep.sh "72 96" "4" "--spp 2048 --steps 1" >> gpurun_out/r5_c12.log 2>&1
cat gpurun_out/r5_c12.log
