cd $GRAFT_REPO_ROOT
SPP=256 VARIANTS="FH_HINT=0;FH_HINT=1;FH_HINT=1,FH_HINT_MARGIN=0.02;FH_HINT=1,FH_HINT_MARGIN=0.08" timeout -k 10 900 python tools/sah_compare.py soup > gpurun_out/r5_hint1.log 2>&1
SPP=64 VARIANTS="FH_HINT=0;FH_HINT=1;FH_HINT=1,FH_HINT_MARGIN=0.02" timeout -k 10 900 python tools/sah_compare.py sponza city >> gpurun_out/r5_hint1.log 2>&1
grep "^soup\|^city\|^sponza" gpurun_out/r5_hint1.log | sed 's/FH_SAH_ITERS=default  builder=auto : build [0-9. ms(call)]*, //; s/, wave steps.*crc/ crc/' | cut -c1-300
