cd $GRAFT_REPO_ROOT
SPP=64 VARIANTS="FH_BOTTOM_UP=0,FH_NO_ALPHA=1;FH_BOTTOM_UP=1,FH_NO_ALPHA=1;FH_BOTTOM_UP=0,FH_NO_ALPHA=1;FH_BOTTOM_UP=1,FH_NO_ALPHA=1" timeout -k 10 900 python tools/sah_compare.py sponza > gpurun_out/r5_bu5.log 2>&1
SPP=256 VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=1;FH_BOTTOM_UP=0;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py soup4 >> gpurun_out/r5_bu5.log 2>&1
grep "^soup\|^city\|^sponza" gpurun_out/r5_bu5.log | sed 's/FH_SAH_ITERS=default  builder=auto : build [0-9. ms(call)]*, //' | cut -c1-400
