"""configs[3] (Sponza-class glTF) under the traversal counters: rays, nodes / triangles per ray and lane utilisation per kernel, then kernel times alone on the GPU.
Developer tool (GPU box): python tools/sponza_probe.py [spp]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fredholm_amd as F
from fredholm_amd import native as N

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = int(os.environ.get("CFG", "3"))
with tempfile.TemporaryDirectory() as td:
    w = bench.workload(cfg, td)
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
bench.apply_environment(r, w)
W, H = w["width"], w["height"]
r.set_resolution(W, H)
L = F.RenderLayer(r, W, H)
cam = F.Camera(**w["camera"])
r.set_flags(N.FLAG_COUNT_TRAVERSAL)
r.reset_stats(); L.clear(); r.init_render_states()
r.render(cam, w["bg"], L, 16, w["depth"]); r.wait_for_completion()
s = r.stats()
for k in ("closest", "shadow"):
    rays = s["rays_" + k]; n = s["nodes_" + k]; t = s["tris_" + k]; wn = s["wave_node_steps_" + k]; wt = s["wave_tri_steps_" + k]
    print(f"{k}: rays {rays} nodes/ray {n/max(rays,1):.2f} tris/ray {t/max(rays,1):.2f} node lane util {n/max(64*wn,1):.3f} tri lane util {t/max(64*wt,1):.3f} "
          f"wave node steps/ray {wn/max(rays,1):.3f} wave tri steps/ray {wt/max(rays,1):.3f}", flush=True)
print("bvh depth", s.get("bvh_depth"), "shaded hits", s.get("shaded_hits"))
for flags, name in ((N.FLAG_SERIAL_PASSES | N.FLAG_TIME_KERNELS, "alone"), (0, "pipelined")):
    r.set_flags(flags)
    L.clear(); r.init_render_states()
    r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()   # warm-up (adaptive tail depth, pools)
    r.reset_stats(); L.clear(); r.init_render_states()
    t0 = time.perf_counter()
    r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
    dt = time.perf_counter() - t0
    s = r.stats()
    print(f"{name}: {W*H*spp/dt/1e6:.1f} Msamples/s ({dt*1e3:.1f} ms for {spp} spp); ms:", {k: round(s[k], 2) for k in s if k.endswith("_ms")}, flush=True)
