"""Emulated strong scaling on ONE GPU: the whole frame, then EACH of the 8 tile shards of an 8-way split rendered in turn (no gather; scene + BVH replicated as on 8 GPUs).
The speed-up of the render phase is whole / max over shards; the gather (~0.6 ms at 1080p) and rank 0's un-permutation come on top.  One JSON line per case on stdout.
    python tools/shard_time.py            env: CONFIGS="2,3" SPPS="1024,16" WORLD=8 TILES="32,16" STEPS=4"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import fredholm_amd as F

WORLD = int(os.environ.get("WORLD", "8"))
STEPS = int(os.environ.get("STEPS", "4"))
tmp = tempfile.TemporaryDirectory()
for cfg in [int(c) for c in os.environ.get("CONFIGS", "2,3").split(",")]:
    w = bench.workload(cfg, tmp.name)
    W, H, D = w["width"], w["height"], w["depth"]
    r = F.Renderer(0); r.load_scene(w["scene"]); r.build_ias()
    bench.apply_environment(r, w)
    if w["sun"] is not None and not w["dir_le"]:
        r.clear_directional_light()
    r.set_resolution(W, H)
    L = F.RenderLayer(r, W, H)
    cam = F.Camera(**w["camera"])

    def timed(spp):
        pool_spp, _, _ = bench.pass_size(r, torch, 0, r.owned_pixel_count(), spp)
        r.set_path_pool(max(int(r.owned_pixel_count() * pool_spp), 1))
        for _ in range(2):
            r.render(cam, w["bg"], L, spp, D); r.wait_for_completion()
        ts = []
        for _ in range(STEPS):
            t0 = time.perf_counter(); r.render(cam, w["bg"], L, spp, D); r.wait_for_completion(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    for spp in [int(s) for s in os.environ.get("SPPS", "1024,16").split(",")]:
        spp = min(spp, 512) if cfg == 3 and spp > 512 else spp
        r.set_tile_shard(0, 1, 32, 32)
        whole = timed(spp)
        for tile in [int(t) for t in os.environ.get("TILES", "32,16").split(",")]:
            shards = []
            for k in range(WORLD):
                r.set_tile_shard(k, WORLD, tile, tile)
                shards.append(round(timed(spp), 3))
            mx, mean = max(shards), sum(shards) / len(shards)
            print(json.dumps({"config": cfg, "workload": w["name"][:40], "spp": spp, "world": WORLD, "tile": tile, "whole_ms": round(whole, 3), "shard_ms": shards, "max_ms": mx, "mean_ms": round(mean, 3),
                              "max_over_mean": round(mx / mean, 4), "render_speedup_whole_over_max": round(whole / mx, 3), "efficiency": round(whole / (WORLD * mx), 4),
                              "note": "one GPU renders each rank's shard in turn (emulated): render phase only, no gather"}), flush=True)
    r.close()
