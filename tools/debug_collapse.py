import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import fredholm_amd as F
from fredholm_amd import scenes
from oracle import pyoracle as O
sc = scenes.cornell_box()
v = sc["vertices"].copy(); v[3:6] = v[3]; sc["vertices"] = v
r = F.Renderer(0); r.load_scene(sc); r.build_ias()
S = O.Scene(sc)
rng = np.random.default_rng(0)
n = 20000
o = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d.astype(np.float32), np.full((n, 1), 1e9, np.float32)], axis=1).astype(np.float32)
rays[:, 1] += 1.0; rays[:500, 1] = 0.0
tg, pg = r.trace_rays(rays); to, po = S.trace(rays)
bad = np.nonzero(pg != po)[0]
print("mismatches", len(bad), "of", n, "stats", r.stats()["bvh_nodes"], r.stats()["bvh_depth"])
for i in bad[:10]:
    print(i, rays[i], "gpu", pg[i], tg[i], "oracle", po[i], to[i])
tb, pb = S.trace(rays, brute=True)
print("oracle vs brute mismatches", int((pb != po).sum()), "gpu vs brute", int((pb != pg).sum()))
