cd $GRAFT_REPO_ROOT
# soak on the final build: the same frame again and again, bit-identical every time -- the soup with the start of its first-hit rays forced to the face's node, decided by
# the library, and the Sponza-class interior
for v in "FH_BOTTOM_UP=1 SCENE=soup" "FH_BOTTOM_UP=2 SCENE=soup" "SCENE=sponza"; do
  echo "== $v"; env $v SOAK_FRAMES=100 timeout -k 10 500 python tools/soak.py 2>&1 | tail -2
done > gpurun_out/r5_soak.log 2>&1; cat gpurun_out/r5_soak.log
