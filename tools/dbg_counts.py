import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import fredholm_amd as F
from fredholm_amd import native as N, scenes
sc = scenes.triangle_soup(1_000_000)
r = F.Renderer(0); r.load_scene(sc); r.build_ias()
r.set_directional_light((0,0,0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0,0.3)
r.set_resolution(1920,1080)
r.set_path_pool(1920*1080*64)
L = F.RenderLayer(r,1920,1080)
cam = F.Camera(**scenes.SOUP_CAMERA)
for spp in (64, 256):
    for rep in range(3):
        r.set_flags(N.FLAG_COUNT_TRAVERSAL); r.reset_stats()
        r.render(cam,(0,0,0),L,spp,8); r.wait_for_completion()
        s = r.stats()
        print(spp, rep, {k: s[k] for k in ("rays_closest","rays_shadow","nodes_shadow","n_closest_launches","n_shadow_launches","paths")}, flush=True)
