import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import native as N, scenes
sc = scenes.triangle_soup(1_000_000)
r = F.Renderer(0); r.load_scene(sc); r.build_ias()
r.set_directional_light((0,0,0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0,0.3)
r.set_resolution(1920,1080)
L = F.RenderLayer(r,1920,1080)
cam = F.Camera(**scenes.SOUP_CAMERA)
r.set_flags(N.FLAG_COUNT_TRAVERSAL)
for depth in (1, 2, 8):
    r.reset_stats(); L.clear(); r.init_render_states()
    r.render(cam,(0,0,0),L,int(os.environ.get("SPP","16")),depth); r.wait_for_completion()
    s = r.stats()
    for k in ("closest","shadow"):
        rays=s["rays_"+k]; n=s["nodes_"+k]; t=s["tris_"+k]; wn=s["wave_node_steps_"+k]; wt=s["wave_tri_steps_"+k]
        print(f"depth<={depth} {k}: rays {rays} nodes/ray {n/max(rays,1):.2f} tris/ray {t/max(rays,1):.2f} node SIMD eff {n/max(64*wn,1):.3f} tri SIMD eff {t/max(64*wt,1):.3f} wave node steps {wn} wave tri steps {wt}")
