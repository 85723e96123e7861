cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/sah_compare.py soup sponza city > gpurun_out/r5_sah1.log 2>&1
