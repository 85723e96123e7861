#!/bin/bash
# same-box A/B of an environment switch on the 1-spp / 16-spp frame times: tools/gpu_latency_ab.sh "<VAR=value>" "<configs>"   (off, on, off, on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "FH_AB_OFF=1" "$1"; do
    echo "== $v"
    env $v timeout -k 10 300 python3 tools/latency_breakdown.py $2 2> gpurun_out/lat.err | cut -c1-200 || { echo FAILED; tail -3 gpurun_out/lat.err; }
  done
done
