#!/bin/bash
# round 4, call 2: parity suite on the new defaults (7 workgroups per CU, scan hand-over), A/B of closest-hit at 8 workgroups and of the emitter variant of the secondary kernel at 6 / 7, the default bench line with the general_scene block
cd $GRAFT_REPO_ROOT
echo "== parity suite"; timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
echo "== configs[3], 512 spp"; bash tools/gpu_ab.sh "old base c8" "3" "--spp 512 --steps 2 --warmup 1 --no-extras"
echo "== configs[2]"; bash tools/gpu_ab.sh "old base c8" "2" "--steps 6 --warmup 2 --no-extras"
echo "== configs[1]"; bash tools/gpu_ab.sh "old base h6 h7" "1" "--steps 4 --warmup 1 --no-extras"
echo "== configs[4], 1024 spp"; bash tools/gpu_ab.sh "base h6 h7" "4" "--spp 1024 --steps 2 --warmup 1 --no-extras"
echo "== default bench line"; timeout -k 10 600 python3 bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "rc=$?"; tail -3 gpurun_out/r4_bench_default.err; cut -c1-300 gpurun_out/r4_bench_default.json
