"""LBVH (radix tree) vs PLOC as the input of the 8-wide collapse: build time, nodes visited per ray and frame time on the uniform
triangle soup and on the non-uniform `city` scene."""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import numpy as np
    import fredholm_amd as F
    from fredholm_amd import native as N, scenes
    name = sys.argv[1]
    sc, camkw = (scenes.triangle_soup(1_000_000), scenes.SOUP_CAMERA) if name == "soup" else (scenes.city(int(os.environ.get("CITY_BLOCKS", "80000"))), scenes.CITY_CAMERA)
    r = F.Renderer(0); r.load_scene(sc); r.build_ias()
    r.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0, 0.3)
    W, H = 1920, 1080
    r.set_resolution(W, H); r.set_path_pool(W * H * 32)
    L = F.RenderLayer(r, W, H); cam = F.Camera(**camkw)
    for _ in range(2): r.render(cam, (0, 0, 0), L, 64, 8)
    r.wait_for_completion()
    t0 = time.perf_counter()
    for _ in range(4): r.render(cam, (0, 0, 0), L, 64, 8)
    r.wait_for_completion()
    dt = (time.perf_counter() - t0) / 4
    r.set_flags(N.FLAG_COUNT_TRAVERSAL); r.reset_stats(); r.render(cam, (0, 0, 0), L, 8, 8); r.wait_for_completion(); s = r.stats()
    b = L.download("beauty")
    print(f"{name:5s} {os.environ.get('FH_BVH_BUILDER', 'lbvh'):5s}: build {s['bvh_build_ms']:.1f} ms, {s['bvh_nodes']} nodes, closest {s['nodes_closest'] / max(s['rays_closest'], 1):.2f} nodes/ray "
          f"{s['tris_closest'] / max(s['rays_closest'], 1):.2f} tris/ray, shadow {s['nodes_shadow'] / max(s['rays_shadow'], 1):.2f} nodes/ray, {dt * 1e3:.1f} ms per 64-spp frame, "
          f"mean {np.nanmean(b[..., :3]):.4f}", flush=True)
else:
    for scene in ("soup", "city"):
        for builder in ("lbvh", "ploc"):
            subprocess.run([sys.executable, __file__, scene], env=dict(os.environ, FH_BVH_BUILDER=builder))
