"""Estimate strong scaling on ONE GPU: time the render of rank 0's tile shard for world sizes 1, 2, 4, 8 (no gather)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fredholm_amd as F
from fredholm_amd import scenes
W, H, SPP, DEPTH = 1920, 1080, int(os.environ.get("SPP", "64")), 8
sc = scenes.triangle_soup(1_000_000)
POOL_SPP = int(os.environ.get('POOL_SPP', '64'))
WORLDS = [int(x) for x in os.environ.get('WORLDS', '1,2,4,8').split(',')]
for world in WORLDS:
    r = F.Renderer(0); r.load_scene(sc); r.build_ias()
    r.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0); r.clear_directional_light(); r.load_arhosek_sky(3.0, 0.3)
    r.set_resolution(W, H)
    if world > 1:
        r.set_tile_shard(0, world, 32, 32)
    if os.environ.get('BENCH_SIZING') == '1':  # passes as bench.py sizes them: equal passes of at most ~128 spp of a 1080p frame, at least three per step
        cap = int(1920 * 1080 * 128 * 1.02)
        passes = max(3, -(-r.owned_pixel_count() * SPP // cap))
        r.set_path_pool(r.owned_pixel_count() * max(-(-SPP // passes), 1))
    else:
        r.set_path_pool(min(r.owned_pixel_count() * POOL_SPP, 160 * 1024 * 1024))
    L = F.RenderLayer(r, W, H)
    cam = F.Camera(**scenes.SOUP_CAMERA)
    for _ in range(3):
        r.render(cam, (0, 0, 0), L, SPP, DEPTH)
    r.wait_for_completion()
    from fredholm_amd import native as N
    r.set_flags(N.FLAG_TIME_KERNELS); r.reset_stats()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        r.render(cam, (0, 0, 0), L, SPP, DEPTH)
        if os.environ.get('STEP_SYNC', '1') == '1':
            r.wait_for_completion()
    r.wait_for_completion()
    dt = (time.perf_counter() - t0) / n
    print(f"world {world}: rank-0 shard {r.owned_pixel_count()} px, {dt*1e3:.2f} ms per step", flush=True)
    st = r.stats()
    print('    ', {k: round(v / n, 3) for k, v in st.items() if k.endswith('_ms') and 'bvh' not in k}, {k: v // n for k, v in st.items() if k.startswith('n_')}, flush=True)
    r.close()
