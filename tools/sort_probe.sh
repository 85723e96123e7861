cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bits in 0 3 4 5 6 7; do
  m=sorted; [ $bits = 0 ] && m=random
  BITS=$bits MODE=$m ANY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_b$bits -- python3 $R/tools/sort_probe.py > /dev/null 2>&1
  echo "bits=$bits: $(find $R/gpurun_out/sp_b$bits -name '*kernel_stats.csv' -exec grep k_trace_batch {} \; | awk -F, '{print $(NF-5)}')"
  rm -rf $R/gpurun_out/sp_b$bits
done
