"""First-light script for the GPU box: KATs + trace parity + tiny render parity, prints a summary."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import fredholm_amd as F
from fredholm_amd import native as N, scenes
from oracle import pyoracle as O

r = F.Renderer(0)
rng = np.random.default_rng(1)
# hashes
inp = rng.integers(0, 2**32, size=(1000, 4), dtype=np.uint32)
out = np.zeros(1000, np.uint32)
N.check(r._ctx, N.lib().fh_kat_hash(r._ctx, 2, 1000, N.ptr(inp), N.ptr(out)), "kat")
ref = np.array([O.xxhash32(*row) for row in inp], dtype=np.uint32)
print("xxhash32x4 exact:", np.array_equal(out, ref))
# trace parity on cornell
sc = scenes.cornell_box()
r.load_scene(sc); r.build_ias()
S = O.Scene(sc)
n = 20000
o = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32); o[:, 1] += 1.0
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.full((n, 1), 1e9, np.float32)], axis=1).astype(np.float32)
t0 = time.time(); tuv_g, prim_g = r.trace_rays(rays); t1 = time.time()
tuv_o, prim_o = S.trace(rays)
print("cornell closest prim equal:", np.array_equal(prim_g, prim_o), "tuv bit-equal:", np.array_equal(tuv_g.view(np.uint32), tuv_o.view(np.uint32)), "hits", (prim_g != 0xffffffff).mean())
# render parity
cam = F.Camera(**scenes.CORNELL_CAMERA)
W = H = 64
r.set_resolution(W, H)
L = F.RenderLayer(r, W, H)
for spp in range(4):
    r.render(cam, (0, 0, 0), L, 1, 5)
r.wait_for_completion()
g = L.download("beauty")
Lo = S.new_layers(W, H)
for spp in range(4):
    S.render(cam.params(), W, H, Lo, 1, 5, n_threads=8)
diff = np.abs(g - Lo["beauty"])
print("cornell render: bit-equal pixels", (g.view(np.uint32) == Lo["beauty"].view(np.uint32)).all(axis=2).mean(), "max abs diff", diff.max(), "mean gpu", g[..., :3].mean(), "mean oracle", Lo["beauty"][..., :3].mean())
for name in ("position", "normal", "albedo", "texcoord", "depth"):
    a, b = L.download(name), Lo[name]
    print(" aov", name, "max diff", np.abs(a - b).max())
# soup
sc2 = scenes.triangle_soup(20000, 0.1)
r2 = F.Renderer(0)
r2.load_scene(sc2); r2.build_ias()
r2.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r2.clear_directional_light()
r2.load_arhosek_sky(3.0, 0.3)
S2 = O.Scene(sc2)
S2.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0)
import ctypes
O.lib().orc_set_directional_light(S2.h, 0, None, None, ctypes.c_float(0))
S2.load_arhosek_sky(3.0, 0.3)
o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
rays = np.concatenate([o, d, np.full((n, 1), 1e9, np.float32)], axis=1).astype(np.float32)
tuv_g, prim_g = r2.trace_rays(rays)
tuv_o, prim_o = S2.trace(rays)
print("soup closest prim equal:", np.array_equal(prim_g, prim_o), "mismatch", (prim_g != prim_o).sum(), "hits", (prim_g != 0xffffffff).mean())
cam2 = F.Camera(**scenes.SOUP_CAMERA)
r2.set_resolution(W, H)
L2 = F.RenderLayer(r2, W, H)
for spp in range(2):
    r2.render(cam2, (0, 0, 0), L2, 1, 8)
r2.wait_for_completion()
g2 = L2.download("beauty")
Lo2 = S2.new_layers(W, H)
for spp in range(2):
    S2.render(cam2.params(), W, H, Lo2, 1, 8, n_threads=8)
print("soup render: bit-equal pixels", (g2.view(np.uint32) == Lo2["beauty"].view(np.uint32)).all(axis=2).mean(), "max abs diff", np.nanmax(np.abs(g2 - Lo2["beauty"])), "mean gpu", g2[..., :3].mean(), "oracle", Lo2["beauty"][..., :3].mean())
# perf probe: 1M soup 1080p
t = time.time(); sc3 = scenes.triangle_soup(1_000_000); print("soup gen s", time.time() - t)
r3 = F.Renderer(0)
t = time.time(); r3.load_scene(sc3); print("upload s", time.time() - t)
t = time.time(); r3.build_ias(); print("bvh build s", time.time() - t, r3.stats()["bvh_build_ms"])
r3.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r3.clear_directional_light()
r3.load_arhosek_sky(3.0, 0.3)
r3.set_resolution(1920, 1080)
L3 = F.RenderLayer(r3, 1920, 1080)
r3.set_flags(N.FLAG_TIME_KERNELS)
r3.render(cam2, (0, 0, 0), L3, 2, 8); r3.wait_for_completion()
r3.reset_stats()
t = time.time(); r3.render(cam2, (0, 0, 0), L3, 8, 8); r3.wait_for_completion(); dt = time.time() - t
st = r3.stats()
print("1080p 8spp depth8: %.3f s => %.2f Msamples/s" % (dt, 1920 * 1080 * 8 / dt / 1e6), {k: round(v, 2) if isinstance(v, float) else v for k, v in st.items()})
