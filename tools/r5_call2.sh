cd $GRAFT_REPO_ROOT
VARIANTS="FH_SAH_ITERS=0;FH_SAH_ITERS=40,FH_SAH_MIN_GAIN=0.0005" timeout -k 10 900 python tools/sah_compare.py soup sponza > gpurun_out/r5_sah2.log 2>&1
