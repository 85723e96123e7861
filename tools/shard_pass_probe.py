"""How should ONE rank of an 8-way split cut its frame into passes?  Rank 0's shard of configs[2] (1024 spp) and configs[3] (512 spp) with the call cut into 1 ... 24 passes
(fh_set_path_pool: the pool decides the cut), median of STEPS frames; the whole frame / 8 is the ideal.  python tools/shard_pass_probe.py   (GPU box)"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import fredholm_amd as F

WORLD, STEPS = 8, int(os.environ.get("STEPS", "5"))
tmp = tempfile.TemporaryDirectory()
for cfg, spp in ((2, 1024), (3, 512)):
    w = bench.workload(cfg, tmp.name)
    W, H, D = w["width"], w["height"], w["depth"]
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
    bench.apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()
    r.set_resolution(W, H)
    L = F.RenderLayer(r, W, H)
    cam = F.Camera(**w["camera"])
    r.set_tile_shard(0, WORLD, 32, 32)
    n = r.owned_pixel_count()

    def timed():
        for _ in range(2):
            r.render(cam, w["bg"], L, spp, D); r.wait_for_completion()
        r.reset_stats()
        ts = []
        for _ in range(STEPS):
            t0 = time.perf_counter(); r.render(cam, w["bg"], L, spp, D); r.wait_for_completion(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2], r.stats()["n_passes"] / STEPS

    default_spp, _, _ = bench.pass_size(r, torch, 0, n, spp)
    r.set_path_pool(max(int(n * default_spp), 1))
    t, p = timed()
    print(json.dumps({"config": cfg, "spp": spp, "pool": "bench default", "pool_spp_per_owned_pixel": round(default_spp, 2), "passes": p, "shard_ms": round(t, 3)}), flush=True)
    for k in (1, 2, 3, 4, 6, 9, 12, 24):
        r.set_path_pool(max(int(n * spp / k) + 1, 1))
        t, p = timed()
        print(json.dumps({"config": cfg, "spp": spp, "pool": f"spp / {k}", "passes": p, "shard_ms": round(t, 3)}), flush=True)
    r.close()
