cd $GRAFT_REPO_ROOT
# EXPERIMENT (tools/ab/lib_exp.so: the counting kernels count only the node visits that found nothing -- no child entered, no triangle to test): how many visits per ray
# a distance kept with the stacked groups could cull when they are popped.  Beside the ordinary counts of the product library.
for lib in "" tools/ab/lib_exp.so; do
  for sc in soup sponza; do
    FH_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} FH_BOTTOM_UP=0 SPP=32 timeout -k 10 300 python tools/sah_compare.py --one $sc 2>/dev/null | sed "s|^|${lib:-product}: |"
  done
done > gpurun_out/r5_empty_visits.log 2>&1; cat gpurun_out/r5_empty_visits.log | cut -c1-420
