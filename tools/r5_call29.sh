cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "opacity or cut or alpha or wild or textur or ring" > gpurun_out/r5_c29_tests.log 2>&1; tail -12 gpurun_out/r5_c29_tests.log
FH_DEBUG_BVH=1 timeout -k 10 300 python - 2>&1 <<'PY' | grep -a "alpha\|counts"
import tempfile, bench
import fredholm_amd as F
w = bench.workload(3, tempfile.mkdtemp())
r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
print("face counts", r.alpha_face_counts(), "cell counts", r.alpha_cell_counts())
r.close()
PY
bash tools/gpu_env_ab3.sh "FH_OPACITY_MICROMAP=0" "3" "--spp 540 --steps 2" 2>&1 | cut -c1-200
