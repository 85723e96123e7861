import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fredholm_amd as F
from fredholm_amd import native as N, scenes
sc = scenes.triangle_soup(1_000_000)
r = F.Renderer(0); r.load_scene(sc); r.build_ias()
r.set_directional_light((0,0,0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0,0.3)
r.set_resolution(1920,1080)
L = F.RenderLayer(r,1920,1080)
cam = F.Camera(**scenes.SOUP_CAMERA)
r.set_flags(N.FLAG_COUNT_TRAVERSAL)
for depth in (1,2,3,8):
    r.reset_stats(); L.clear(); r.init_render_states()
    r.render(cam,(0,0,0),L,2,depth); r.wait_for_completion()
    s = r.stats()
    print("depth<=%d closest hist" % depth, s["hist_nodes_closest"], "shadow hist", s["hist_nodes_shadow"])
