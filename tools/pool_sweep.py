import sys, os, time
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
for spp_per_pass in (2, 4, 8, 16, 32, 64):
    r.set_path_pool(1920*1080*spp_per_pass)
    spp = max(16, spp_per_pass)
    r.render(cam,(0,0,0),L,spp,8); r.wait_for_completion()
    t=time.perf_counter()
    for _ in range(3): r.render(cam,(0,0,0),L,spp,8)
    r.wait_for_completion(); dt=(time.perf_counter()-t)/3
    print(f"spp/pass {spp_per_pass:3d}: {dt*1e3:8.2f} ms per {spp} spp -> {1920*1080*spp/dt/1e6:8.1f} Msamples/s")
