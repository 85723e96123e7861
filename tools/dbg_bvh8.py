import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import scenes
from oracle import pyoracle as O
for n_tris, edge in [(1,0.5),(2,0.5),(3,0.5),(5,0.5),(64,0.3),(5000,0.1)]:
    sc = scenes.triangle_soup(n_tris, edge)
    r = F.Renderer(0); r.load_scene(sc); r.build_ias()
    S = O.Scene(sc)
    rng = np.random.default_rng(n_tris)
    n=30000
    o = rng.uniform(-1.4,1.4,(n,3)).astype(np.float32)
    d = rng.normal(size=(n,3)).astype(np.float32); d/=np.linalg.norm(d,axis=1,keepdims=True)
    rays = np.concatenate([o,d,np.full((n,1),1e9,np.float32)],axis=1).astype(np.float32)
    v = sc["vertices"].reshape(-1,3,3)
    pick = rng.integers(0,n_tris,2000)
    rays[:2000,0:3] = v[pick].mean(axis=1)
    tg,pg = r.trace_rays(rays); to,po = S.trace(rays)
    bad = np.nonzero(pg!=po)[0]
    print(n_tris, "mismatch", bad.size, "of", n, "first", bad[:5], "gpu", pg[bad[:5]], "orc", po[bad[:5]], "t gpu", tg[bad[:5],0], "t orc", to[bad[:5],0])
    if bad.size:
        i=bad[0]; print(" ray", rays[i])
    r.close()
