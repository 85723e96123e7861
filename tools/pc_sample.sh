#!/bin/bash
# PC sampling of the bench (beta feature of rocprofv3); writes gpurun_out/pcs_<method>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d $R/gpurun_out/pcs_stoch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pcs_stoch.log 2>&1
echo "stochastic rc=$?"; tail -3 $R/gpurun_out/pcs_stoch.log
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --output-format csv -d $R/gpurun_out/pcs_host -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pcs_host.log 2>&1
echo "host_trap rc=$?"; tail -3 $R/gpurun_out/pcs_host.log
ls -la $R/gpurun_out/pcs_*/*/ 2>/dev/null | head -20
