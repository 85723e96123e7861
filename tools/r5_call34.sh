cd $GRAFT_REPO_ROOT
# first-hit rays of scenes WITH cut-outs start at the node of the face they leave (FH_BOTTOM_UP_ALPHA): parity, then configs[3] with the start forced off / on / decided
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cut_out or node_of_their_face or opacity or refit" > gpurun_out/r5_bu_alpha_tests.log 2>&1 || { tail -30 gpurun_out/r5_bu_alpha_tests.log; exit 1; }
tail -2 gpurun_out/r5_bu_alpha_tests.log
: > gpurun_out/r5_bu_alpha_ab.log
for v in 0 1 2 0 1; do
  echo "== FH_BOTTOM_UP=$v" >> gpurun_out/r5_bu_alpha_ab.log
  FH_DEBUG_BVH=1 FH_BOTTOM_UP=$v timeout -k 10 300 python bench.py --config 3 --no-extras --no-cpu-baseline --steps 2 --warmup 1 2> gpurun_out/r5_bu_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'], d.get('kernel_ms_per_step_alone'))" >> gpurun_out/r5_bu_alpha_ab.log || exit 1
  grep "start at" gpurun_out/r5_bu_err.log | tail -1 >> gpurun_out/r5_bu_alpha_ab.log
done
cat gpurun_out/r5_bu_alpha_ab.log
