import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
from oracle import pyoracle as O
import test_gpu_parity as T
import fredholm_amd as F
O.lib()
cap = {}
class Stop(Exception): pass
def fake(oracle, sc, cam, w, h, launches, spp_per_launch, depth, setup=None, bg=(0, 0, 0), pool=None):
    cap.update(sc=sc, cam=cam, setup=setup, bg=bg, w=w, h=h)
    raise Stop
T._render_pair = fake
def render(env, depth, launches=1):
    for k in ("FH_TAIL_STREAM", "FH_TAIL_DEPTH"): os.environ.pop(k, None)
    os.environ.update(env)
    r = F.Renderer(0)
    r.load_scene(cap["sc"]); r.build_ias()
    class Dummy:  # the setup also drives the oracle scene: give it the renderer only
        pass
    cap["setup"](r)
    r.set_resolution(cap["w"], cap["h"])
    L = F.RenderLayer(r, cap["w"], cap["h"])
    for _ in range(launches): r.render(cap["cam"], cap["bg"], L, 1, depth)
    r.wait_for_completion()
    img = L.download("beauty"); st = r.stats(); r.close()
    return img, st
for seed in (3,):
    try: T.test_random_materials_textures_and_lights_match_checker(O, seed)
    except Stop: pass
    m = cap["sc"]["materials"]
    print("base_color", np.asarray(m["base_color"]).tolist())
    print("material ids", cap["sc"]["material_ids"].tolist())
    a, st = render({"FH_TAIL_STREAM": "0", "FH_TAIL_DEPTH": "1"}, 2)
    for dbg in (0, 1, 2, 3, 5, 9, 7, 15):
        b, _ = render({"FH_TAIL_STREAM": "1", "FH_TAIL_DEPTH": "1", "FH_TAIL_DBG": str(dbg)}, 2)
        d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
        ys, xs = np.nonzero(d)
        print("dbg", dbg, int(d.sum()), [(int(y) * 48 + int(x), float(b[y, x, 0] / a[y, x, 0])) for y, x in zip(ys, xs)], flush=True)
