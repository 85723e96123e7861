cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "opacity or cut or alpha or wild or textur" > gpurun_out/r5_c10_tests.log 2>&1; tail -15 gpurun_out/r5_c10_tests.log
FH_DEBUG_BVH=1 timeout -k 10 300 python - > gpurun_out/r5_c10_sponza.log 2>&1 <<'PY'
import tempfile, bench
import fredholm_amd as F
w = bench.workload(3, tempfile.mkdtemp())
r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
print("alpha face counts (can cut, always pass, never pass, still tested):", r.alpha_face_counts())
r.close()
PY
grep -a "alpha" gpurun_out/r5_c10_sponza.log
