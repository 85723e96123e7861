"""closest / secondary kernel times on triangle soups of several sizes (developer tool, GPU box): FH_STREAM_CHUNK=<n> python tools/chunk_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fredholm_amd as F
from fredholm_amd import native as N, scenes
for n_tris in (1000, 10000, 100000):
    sc = scenes.triangle_soup(n_tris)
    r = F.Renderer(0); r.load_scene(sc); r.build_ias()
    r.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r.load_arhosek_sky(3.0, 0.3)
    r.set_resolution(1920, 1080)
    L = F.RenderLayer(r, 1920, 1080)
    cam = F.Camera(**scenes.SOUP_CAMERA)
    r.set_flags(N.FLAG_SERIAL_PASSES | N.FLAG_TIME_KERNELS)
    for rep in range(2):
        r.reset_stats(); L.clear(); r.init_render_states()
        r.render(cam, (0, 0, 0), L, 256, 8); r.wait_for_completion()
    s = r.stats()
    print(f"{n_tris} triangles, chunk {os.environ.get('FH_STREAM_CHUNK', 'default')}: render {s['render_ms']:.1f} ms closest {s['trace_closest_ms']:.2f} secondary {s['trace_shadow_ms']:.2f} shade {s['shade_ms']:.2f}", flush=True)
    r.close()
