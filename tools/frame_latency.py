"""Frame time of small launches (the reference's GUI renders 1 spp per frame, app/controller.cpp): python tools/frame_latency.py  (GPU box)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fredholm_amd as F
for cfg in (1, 2):
    with tempfile.TemporaryDirectory() as td:
        w = bench.workload(cfg, td)
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
    bench.apply_environment(r, w)
    for (W, H) in ((1920, 1080), (512, 512)):
        r.set_resolution(W, H)
        L = F.RenderLayer(r, W, H)
        cam = F.Camera(**w["camera"])
        for spp in (1, 4):
            for _ in range(20):
                r.render(cam, w["bg"], L, spp, w["depth"])
            r.wait_for_completion()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
            dt = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                r.render(cam, w["bg"], L, spp, w["depth"])
            r.wait_for_completion()
            dq = (time.perf_counter() - t0) / n
            print(f"config {cfg} {W}x{H} {spp} spp per call: {dt*1e3:.3f} ms per frame with a sync per frame, {dq*1e3:.3f} ms queued back to back ({W*H*spp/dq/1e6:.0f} Msamples/s)", flush=True)
    r.close()
