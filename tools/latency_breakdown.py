"""Where the time of a small launch goes (the reference's GUI renders 1 spp per frame, app/controller.cpp:224; rtcamp8 16, rtcamp8.cpp:183-189):
per-kernel-family HIP-event times of 1-spp and 16-spp 1080p frames.  python tools/latency_breakdown.py [config ...]   (GPU box)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fredholm_amd as F
from fredholm_amd import native as N
cfgs = [int(a) for a in sys.argv[1:]] or [1, 2]
for cfg in cfgs:
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
    for spp in (1, 16):
        for _ in range(30):
            r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
        n = 100
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion(); ts.append(time.perf_counter() - t0)
        ts.sort()
        r.set_flags(N.FLAG_TIME_KERNELS); r.reset_stats()
        for _ in range(n):
            r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
        s = r.stats(); r.set_flags(0)
        k = {a: s[a + "_ms"] / n for a in ("generate", "trace_closest", "queue", "shade", "trace_shadow", "tail", "accumulate", "render")}
        print(f"config {cfg} {W}x{H} {spp} spp: median {ts[n // 2] * 1e3:.3f} ms (min {ts[0] * 1e3:.3f}); with events: " + " ".join(f"{a} {v:.3f}" for a, v in k.items()) +
              f" | launches per frame: closest {s['n_closest_launches'] / n:.1f} shade {s['n_shade_launches'] / n:.1f} tail {s['n_tail_launches'] / n:.1f}", flush=True)
    r.close()
