cd $GRAFT_REPO_ROOT
SPP=256 VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=1;FH_BOTTOM_UP=0;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py soup > gpurun_out/r5_bu4.log 2>&1
SPP=64 VARIANTS="FH_BOTTOM_UP=0;FH_BOTTOM_UP=1" timeout -k 10 900 python tools/sah_compare.py city >> gpurun_out/r5_bu4.log 2>&1
grep "^soup\|^city" gpurun_out/r5_bu4.log | sed 's/FH_SAH_ITERS=default  builder=auto : build [0-9. ms(call)]*, //' | cut -c1-400
