"""fh_render(n) + fh_sync for n = 1, 4, 16 at 1080p on configs[1..3] (the reference's own call pattern): median / min ms, and a CRC of the frame.
    python tools/latency_small_calls.py [cfg ...]      env: any FH_* switch"""
import os, sys, time, tempfile, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import fredholm_amd as F
for cfg in [int(a) for a in sys.argv[1:]] or [2, 3, 1]:
    with tempfile.TemporaryDirectory() as td:
        w = bench.workload(cfg, td)
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
    bench.apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()
    W, H = 1920, 1080
    r.set_resolution(W, H)
    L = F.RenderLayer(r, W, H)
    cam = F.Camera(**w["camera"])
    out = []
    for spp, n in ((1, 200), (4, 100), (16, 60)):
        for _ in range(20):
            r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out.append(f"{spp} spp {ts[len(ts) // 2]:.3f} / {ts[0]:.3f} ms")
    L.clear(); r.init_render_states()
    for spp in (1, 1, 16):
        r.render(cam, w["bg"], L, spp, w["depth"])
    r.wait_for_completion()
    crc = zlib.crc32(np.ascontiguousarray(L.download("beauty")).tobytes())
    print(f"configs[{cfg}]: " + ", ".join(out) + f" (median / min), crc of 1+1+16 spp {crc:08x}", flush=True)
    r.close()
