cd $GRAFT_REPO_ROOT
VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py soup sponza > gpurun_out/r5_bu2.log 2>&1
