"""Does spatial ordering of incoherent rays pay?  Trace the same any-hit rays through fh_trace_rays in random order and sorted by
the Morton code of their origin; kernel times come from rocprofv3 --kernel-trace --stats of this script."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import scenes
sc = scenes.triangle_soup(1_000_000)
r = F.Renderer(0); r.load_scene(sc); r.build_ias()
rng = np.random.default_rng(5)
n = 4_000_000
o = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.full((n, 1), 1e9, np.float32)], axis=1).astype(np.float32)
def morton(p, bits):
    q = np.clip(((p + 1) * 0.5 * (1 << bits)).astype(np.uint32), 0, (1 << bits) - 1)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + a)
    return code
order = np.argsort(morton(o, int(os.environ.get("BITS", "7"))), kind="stable")
mode = os.environ.get("MODE", "random")
batch = rays if mode == "random" else np.ascontiguousarray(rays[order])
for _ in range(3):
    tuv, prim = r.trace_rays(batch, any_hit=(os.environ.get("ANY", "1") == "1"))
print(mode, "hits", (prim != 0xFFFFFFFF).mean())
