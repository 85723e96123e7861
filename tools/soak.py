"""Soak: render the same 1080p frame many times (two passes in flight, cell-ordered queues, adaptive switch depth) and require the
beauty layer to be bit-identical every time."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import scenes
N = int(os.environ.get("SOAK_FRAMES", "150"))
CITY = os.environ.get("SCENE", "soup") == "city"
SPONZA = os.environ.get("SCENE", "soup") == "sponza"  # the Sponza-class interior of configs[3]: cut-outs (parked any-hit tests), textures, a 15-level tree (spilled stack)
if SPONZA:
    import tempfile
    import bench
    with tempfile.TemporaryDirectory() as td:
        wl = bench.workload(3, td)
        r = F.Renderer(0); r.load_scene(wl["scene"]); r.build_ias()
else:
    r = F.Renderer(0); r.load_scene(scenes.city(80000) if CITY else scenes.triangle_soup(1_000_000)); r.build_ias()
r.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0, 0.3)
W, H = 1920, 1080
r.set_resolution(W, H)
r.set_path_pool(W * H * 8)   # 4 passes of 8 spp per frame: both streams and both pools busy
L = F.RenderLayer(r, W, H)
cam = F.Camera(**(wl["camera"] if SPONZA else (scenes.CITY_CAMERA if CITY else scenes.SOUP_CAMERA)))
ref = None
t0 = time.perf_counter()
for k in range(N):
    L.clear(); r.init_render_states()
    r.render(cam, (0, 0, 0), L, 32, 8); r.wait_for_completion()
    h = hashlib.sha1(L.download("beauty").tobytes()).hexdigest()
    if ref is None: ref = h
    if h != ref:
        print("MISMATCH at frame", k, h, ref); sys.exit(1)
print(f"soak ok: {N} frames identical ({ref[:12]}), {time.perf_counter() - t0:.1f} s")
