"""SAH reinsertion on / off (FH_SAH_ITERS): build time, summed inner-node area, nodes and triangles per closest-hit / secondary ray, frame throughput
on configs[2] (soup), configs[3] (Sponza-class interior) and scenes.city.    python tools/sah_compare.py [scene ...]   (children: python tools/sah_compare.py --one <scene>)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import numpy as np
    import bench
    import fredholm_amd as F
    from fredholm_amd import native as N, scenes
    name = sys.argv[2]
    tmp = tempfile.TemporaryDirectory()
    if name == "city":
        w = dict(scene=scenes.city(int(os.environ.get("CITY_BLOCKS", "80000"))), camera=scenes.CITY_CAMERA, sky=(3.0, 0.3), sun=scenes.SOUP_SUN, dir_le=None, bg=(0, 0, 0), depth=8, width=1920, height=1080)
    else:
        w = bench.workload({"soup": 2, "sponza": 3, "cornell": 1, "soup4": 4}[name], tmp.name)
    r = F.Renderer(0); r.load_scene(w["scene"])
    t0 = time.perf_counter(); r.build_ias(); t_build = (time.perf_counter() - t0) * 1e3
    bench.apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()
    W, H = 1920, 1080
    spp = int(os.environ.get("SPP", "64"))
    r.set_resolution(W, H); r.set_path_pool(W * H * 32)
    L = F.RenderLayer(r, W, H); cam = F.Camera(**w["camera"])
    for _ in range(2): r.render(cam, w["bg"], L, spp, w["depth"])
    r.wait_for_completion()
    t0 = time.perf_counter()
    for _ in range(4): r.render(cam, w["bg"], L, spp, w["depth"])
    r.wait_for_completion()
    dt = (time.perf_counter() - t0) / 4
    r.set_flags(N.FLAG_TIME_KERNELS | N.FLAG_SERIAL_PASSES); r.reset_stats(); r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion(); a = r.stats()
    r.set_flags(N.FLAG_COUNT_TRAVERSAL); r.reset_stats(); r.render(cam, w["bg"], L, 8, w["depth"]); r.wait_for_completion(); s = r.stats()
    r.set_flags(0)
    b = L.download("beauty")
    import zlib
    print(f"{name:7s} FH_SAH_ITERS={os.environ.get('FH_SAH_ITERS', 'default'):8s} builder={os.environ.get('FH_BVH_BUILDER', 'auto'):5s}: build {s['bvh_build_ms']:.1f} ms (call {t_build:.1f}), {s['bvh_nodes']} nodes, depth {s['bvh_depth']}, "
          f"closest {s['nodes_closest'] / max(s['rays_closest'], 1):.2f} nodes + {s['tris_closest'] / max(s['rays_closest'], 1):.2f} tris per ray, "
          f"secondary {s['nodes_shadow'] / max(s['rays_shadow'], 1):.2f} + {s['tris_shadow'] / max(s['rays_shadow'], 1):.2f}, "
          f"{W * H * spp / dt / 1e6:.1f} Msamples/s ({dt * 1e3:.1f} ms per {spp}-spp frame; alone: closest {a['trace_closest_ms']:.1f} secondary {a['trace_shadow_ms']:.1f} shade {a['shade_ms']:.1f}), "
          f"wave steps per ray: closest {s['wave_node_steps_closest'] * 64 / max(s['rays_closest'], 1):.2f} node + {s['wave_tri_steps_closest'] * 64 / max(s['rays_closest'], 1):.2f} tri, "
          f"secondary {s['wave_node_steps_shadow'] * 64 / max(s['rays_shadow'], 1):.2f} + {s['wave_tri_steps_shadow'] * 64 / max(s['rays_shadow'], 1):.2f}, "
          f"crc {zlib.crc32(np.ascontiguousarray(b).tobytes()):08x}", flush=True)
    r.close()
else:
    names = sys.argv[1:] or ["soup", "sponza", "city"]
    variants = [v.split(",") for v in os.environ.get("VARIANTS", "FH_SAH_ITERS=0;FH_SAH_ITERS=8").split(";")]
    for scene in names:
        for v in variants:
            env = dict(os.environ, FH_DEBUG_BVH="1")
            env.update(dict(kv.split("=") for kv in v if kv))
            subprocess.run([sys.executable, __file__, "--one", scene], env=env)
