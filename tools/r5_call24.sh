cd $GRAFT_REPO_ROOT
bash tools/gpu_env_ab3.sh "FH_STREAM_CHUNK=128" "2" "--steps 8" 2>&1 | cut -c1-200
bash tools/gpu_env_ab3.sh "FH_STREAM_CHUNK=128" "4" "--spp 2048 --steps 1" 2>&1 | cut -c1-200
bash tools/gpu_env_ab3.sh "FH_STREAM_CHUNK=128" "3" "--spp 540 --steps 2" 2>&1 | cut -c1-200
