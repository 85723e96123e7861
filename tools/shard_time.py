"""Emulated strong scaling on ONE GPU: the whole frame, then EACH of the 8 tile shards of an 8-way split rendered in turn (no gather; scene + BVH replicated as on 8 GPUs).
render_speedup = whole / slowest shard (render phase); step_speedup adds what rank 0 does per presented frame: pack, the priced gather, fh_unpack_shards, the post chain.  One JSON line per case on stdout.
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

    # ---- what rank 0 does per presented frame BESIDES rendering its shard (round 6): its own pack, the un-permutation of all WORLD shards in one launch (fh_unpack_shards)
    # and -- configurations with a post chain -- bloom / aberration / tone map on the assembled frame.  Timed alone on this GPU, median of 20, with the buffers the bench uses;
    # the gather itself (WORLD x 4.15 MB at 1080p into rank 0 over seven xGMI links of ~153 GB/s each, in parallel) is priced, not measured: no second GPU here
    from fredholm_amd import distributed as Dd
    from fredholm_amd.renderer import PostProcessParams
    dev = torch.device("cuda", 0)
    pad = Dd.max_owned(W, H, WORLD)
    beauty = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    packed = [torch.zeros((pad, 4), dtype=torch.float32, device=dev) for _ in range(WORLD)]
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    r.set_tile_shard(0, WORLD, 32, 32)

    def med(fn, n=20):
        for _ in range(3):
            fn(); r.wait_for_completion()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); r.wait_for_completion(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    sink = {"pack_ms": round(med(lambda: r.pack_owned(beauty.data_ptr(), 4, packed[0].data_ptr())), 4),
            "unpack_all_ms": round(med(lambda: r.unpack_shards([t.data_ptr() for t in packed], 4, frame.data_ptr())), 4),
            "gather_ms_priced": round(pad * 16 / 153e9 * 1e3 + 0.02, 4),  # one shard per link, the seven links in parallel, + 20 us of launch latency
            "post_ms": 0.0}
    if w["post"]:
        pp = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
        post = PostProcessParams(**w["post"])
        sink["post_ms"] = round(med(lambda: r.post_process(frame.data_ptr(), pp[0].data_ptr(), pp[1].data_ptr(), W, H, post, pp[2].data_ptr())), 4)
    sink["total_ms"] = round(sink["pack_ms"] + sink["gather_ms_priced"] + sink["unpack_all_ms"] + sink["post_ms"], 4)
    r.set_tile_shard(0, 1, 32, 32)
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
                              "rank0_sink": sink, "whole_step_ms": round(whole + sink["post_ms"], 3), "sharded_step_ms": round(mx + sink["total_ms"], 3),
                              "step_speedup": round((whole + sink["post_ms"]) / (mx + sink["total_ms"]), 3),
                              "note": "one GPU renders each rank's shard in turn (emulated).  render_speedup: render phase only; step_speedup: the presented frame -- slowest shard + rank 0's pack, "
                                      "the PRICED gather, the one-launch un-permutation and the post chain, against the unsharded frame + its post chain"}), flush=True)
    r.close()
