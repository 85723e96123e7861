"""resident workgroups per CU of the streaming kernels for a scene, as the runtime reports them (FH_DEBUG_BVH prints the line): python tools/occupancy_probe.py [config]"""
import os, sys, tempfile
os.environ["FH_DEBUG_BVH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fredholm_amd as F
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with tempfile.TemporaryDirectory() as td:
    w = bench.workload(cfg, td)
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
bench.apply_environment(r, w)
r.set_resolution(640, 360)
L = F.RenderLayer(r, 640, 360)
r.render(F.Camera(**w["camera"]), w["bg"], L, 1, 4); r.wait_for_completion()
print("bvh", r.stats()["bvh_depth"], "levels", r.stats()["bvh_nodes"], "nodes")
