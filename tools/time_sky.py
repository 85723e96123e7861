import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import fredholm_amd as F
from fredholm_amd import native as N, scenes
r = F.Renderer(0)
r.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0, 0.3)
n = 8_000_000
rng = np.random.default_rng(1)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 1] = np.abs(d[:, 1])
out = np.zeros((n, 3), np.float32)
for _ in range(3):
    N.check(r._ctx, N.lib().fh_kat_sky(r._ctx, n, N.ptr(d), N.ptr(out)), "sky")
print("done", out.mean())
