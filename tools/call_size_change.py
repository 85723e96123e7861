"""What the first big call after small calls costs (and the first small calls after a big one): the depth at which the fused tail takes over is adapted from
the survivor counts of EARLIER passes, whatever their size.
    python tools/call_size_change.py [cfg ...]      env: any FH_* switch"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import fredholm_amd as F
BIG = int(os.environ.get("BIG_SPP", "256"))
for cfg in [int(a) for a in sys.argv[1:]] or [3, 2]:
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

    def call(spp):
        r.reset_stats()
        t0 = time.perf_counter(); r.render(cam, w["bg"], L, spp, w["depth"]); r.wait_for_completion()
        return (time.perf_counter() - t0) * 1e3, r.stats().get("n_tail_launches")
    for _ in range(2): call(BIG)                      # pools allocated, depth adapted to big passes
    steady_big = [call(BIG) for _ in range(3)]
    for _ in range(30): call(1)
    steady_small = sorted(call(1)[0] for _ in range(50))[25]
    after_small = [call(BIG) for _ in range(3)]       # the first big calls after small ones
    first_small = [call(1)[0] for _ in range(5)]      # the first small calls after big ones
    fmt = lambda xs: " / ".join(f"{t:.1f} ms ({n} tail launches)" for t, n in xs)
    print(f"configs[{cfg}] {BIG}-spp call: steady {fmt(steady_big)}; first three after thirty 1-spp calls {fmt(after_small)}; 1-spp call: steady {steady_small:.2f} ms, first five after big calls "
          + " / ".join(f"{t:.2f}" for t in first_small), flush=True)
    r.close()
