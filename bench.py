#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the configuration the metric and target are quoted on):
1,000,000-triangle synthetic soup, Hosek sky, 1920x1080, max_depth 8, seed 1.
One step = one fh_render of `--spp` samples per pixel of that frame (default 64 = 1/16 of the configuration's
1024 spp; the path pool is sized so that a step is one pass: 133 M path slots, 50 GB, at N = 1) with every input resident in HBM, plus -- for N > 1 -- the RCCL all_gather of the
packed beauty tiles to every rank.  For N > 1 the frame is sharded by interleaved 32x32 pixel tiles
(one process per GPU), so total work is fixed: strong scaling.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, MAX_DEPTH, N_TRIS = 1920, 1080, 8, 1_000_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
TRI_BYTES, RAY_BYTES, HIT_BYTES = 48, 32, 16  # algorithmic bytes per ray


def cpu_baseline(scene_dict, seconds_target=12.0):
    """Time the CPU checker (oracle/, kind "port") on a bounded sample of the SAME workload: whole rows of the
    1080p frame, 1 spp, depth 8, all host threads.  Reported next to the GPU number; never the thing measured."""
    import numpy as np
    from fredholm_amd import scenes
    from fredholm_amd.renderer import Camera
    from oracle import pyoracle as O

    S = O.Scene(scene_dict)
    S.set_directional_light((0, 0, 0), scenes.SOUP_SUN, 0.0)
    import ctypes
    O.lib().orc_set_directional_light(S.h, 0, None, None, ctypes.c_float(0))
    S.load_arhosek_sky(3.0, 0.3)
    cam = Camera(**scenes.SOUP_CAMERA).params()
    threads = max(1, O.hardware_threads())
    # whole-frame 1-spp passes (the running mean continues across passes) until ~seconds_target of CPU work
    L = S.new_layers(WIDTH, HEIGHT)
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < seconds_target and passes < 64:
        S.render(cam, WIDTH, HEIGHT, L, 1, MAX_DEPTH, n_threads=threads)
        passes += 1
        dt = time.perf_counter() - t0
    return {"value": round(WIDTH * HEIGHT * passes / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{passes} spp of the full 1920x1080 frame, max_depth 8, same 1M-triangle scene ({dt:.1f} s wall on {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per step: one fh_render call = one presented frame of BASELINE configs[2] (1080p, 1024 spp)")
    ap.add_argument("--pool-spp", type=int, default=0, help="samples per pixel per pass of the path pool; 0 = equal passes of at most ~64 spp of the full frame, at least two per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-frame", action="store_true", help="N > 1: rank 0 re-renders the whole frame unsharded and compares it bit for bit with the gathered one")
    args = ap.parse_args()

    import numpy as np
    import torch

    import fredholm_amd as F
    from fredholm_amd import distributed as D
    from fredholm_amd import native as N
    from fredholm_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py` (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    backend = os.environ.get("FH_BENCH_BACKEND", "nccl")  # "gloo": functional test of the N > 1 path on a box with fewer GPUs
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- scene, BVH, environment: everything resident in HBM before the timed region
    sc = scenes.triangle_soup(N_TRIS)
    r = F.Renderer(local_rank)
    r.load_scene(sc)
    r.build_ias()
    r.set_directional_light((0.0, 0.0, 0.0), scenes.SOUP_SUN, 0.0)  # sets the sun direction (renderer.h:563) ...
    r.clear_directional_light()                                      # ... without a directional light
    r.load_arhosek_sky(3.0, 0.3)
    r.set_resolution(WIDTH, HEIGHT)
    if world > 1:
        r.set_tile_shard(rank, world, 32, 32)
    # path-pool slots = owned pixels x samples per pass (384 B per slot, two pools).  By default a step is split into equal passes of
    # at most ~64 spp of the full frame (135 M paths, 52 GB per pool), and into at least two, so that the library's
    # two-passes-in-flight pipelining has something to overlap inside a step
    n_owned_now = r.owned_pixel_count()
    if args.pool_spp > 0:
        pool_spp = args.pool_spp
    else:
        cap = int(WIDTH * HEIGHT * 64 * 1.02)  # 2 % slack: tile ownership is not perfectly even across ranks
        passes = max(2, -(-n_owned_now * args.spp // cap))
        pool_spp = max(-(-args.spp // passes), 1)
    r.set_path_pool(n_owned_now * pool_spp)
    cam = F.Camera(**scenes.SOUP_CAMERA)
    dev = torch.device("cuda", local_rank)
    bufs = {n: torch.zeros((HEIGHT, WIDTH) if n == "depth" else (HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES}
    layers = F.RenderLayer(r, WIDTH, HEIGHT, pointers={n: t.data_ptr() for n, t in bufs.items()})
    n_owned = r.owned_pixel_count()
    pad = D.max_owned(WIDTH, HEIGHT, world) if world > 1 else n_owned
    packed = torch.zeros((pad, 4), dtype=torch.float32, device=dev)
    gathered = [torch.empty_like(packed) for _ in range(world)] if world > 1 else None
    frame = torch.zeros((HEIGHT * WIDTH, 4), dtype=torch.float32, device=dev) if world > 1 else None
    own_idx = [torch.from_numpy(D.tile_ownership(WIDTH, HEIGHT, k, world).astype(np.int64)).to(dev) for k in range(world)] if world > 1 else None
    torch.cuda.synchronize()

    def step():
        r.render(cam, (0.0, 0.0, 0.0), layers, args.spp, MAX_DEPTH)
        if world > 1:
            r.pack_owned(bufs["beauty"].data_ptr(), 4, packed.data_ptr())
            r.wait_for_completion()          # library stream -> host; the collective runs on torch's stream
            if backend == "nccl":
                dist.all_gather(gathered, packed)
            else:                            # test path: stage through host memory
                host = [torch.empty(packed.shape, dtype=packed.dtype) for _ in range(world)]
                dist.all_gather(host, packed.cpu())
                for k in range(world):
                    gathered[k].copy_(host[k])
            for k in range(world):           # present the assembled frame
                frame.index_copy_(0, own_idx[k], gathered[k][: own_idx[k].numel()])
            torch.cuda.synchronize()         # the next step's pack must not overwrite `packed` under the collective

    def fence():
        r.wait_for_completion()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    r.set_flags(N.FLAG_TIME_KERNELS)  # HIP events around the traversal/shade launches on the library's own stream
    r.reset_stats()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timed = r.stats()

    # ---- algorithmic bytes of the dominant traversal kernel: one instrumented (counting) replay of a step, untimed.
    # Deterministic sampling makes every step trace the same number of rays/nodes/triangles up to the sample index.
    r.set_flags(N.FLAG_COUNT_TRAVERSAL)
    r.reset_stats()
    r.render(cam, (0.0, 0.0, 0.0), layers, args.spp, MAX_DEPTH)
    r.wait_for_completion()
    cnt = r.stats()
    r.set_flags(0)

    if args.check_frame and world > 1 and rank == 0:
        total_spp = args.spp * (args.warmup + args.steps + 1)
        r2 = F.Renderer(local_rank)
        r2.load_scene(sc)
        r2.build_ias()
        r2.set_directional_light((0.0, 0.0, 0.0), scenes.SOUP_SUN, 0.0)
        r2.clear_directional_light()
        r2.load_arhosek_sky(3.0, 0.3)
        r2.set_resolution(WIDTH, HEIGHT)
        full = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
        others = {n: torch.zeros((HEIGHT, WIDTH) if n == "depth" else (HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev) for n in F.RenderLayer.NAMES}
        others["beauty"] = full
        layers2 = F.RenderLayer(r2, WIDTH, HEIGHT, pointers={n: t.data_ptr() for n, t in others.items()})
        # 1. the frame assembled from every rank's tiles after the last timed step against an unsharded render of as many samples
        r2.render(cam, (0.0, 0.0, 0.0), layers2, total_spp - args.spp, MAX_DEPTH)
        r2.wait_for_completion()
        whole = bool((full.reshape(-1, 4).view(torch.int32) == frame.view(torch.int32)).all().item())
        print(f"check-frame: frame gathered from {world} ranks bit-identical to the unsharded render: {whole}", file=sys.stderr, flush=True)
        if not whole:
            raise SystemExit("gathered frame differs from the unsharded one")
        # 2. this rank's tiles after the counting replay (one more step that only the local layers saw)
        r2.render(cam, (0.0, 0.0, 0.0), layers2, args.spp, MAX_DEPTH)
        r2.wait_for_completion()
        a = full.reshape(-1, 4).view(torch.int32)
        r.pack_owned(bufs["beauty"].data_ptr(), 4, packed.data_ptr())
        r.wait_for_completion()
        mine = packed[: own_idx[0].numel()].view(torch.int32)
        ok = bool((a[own_idx[0]] == mine).all().item())
        print(f"check-frame: rank-0 shard bit-identical to the unsharded render: {ok}", file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit("sharded render differs from the unsharded one")
        r2.close()
    if rank == 0:
        samples = WIDTH * HEIGHT * args.spp * args.steps
        value = samples / dt / 1e6
        kernels = {
            "k_trace_closest": (timed["trace_closest_ms"], timed["n_closest_launches"], cnt["rays_closest"], cnt["nodes_closest"], cnt["tris_closest"], cnt["n_closest_launches"]),
            "k_trace_secondary": (timed["trace_shadow_ms"], timed["n_shadow_launches"], cnt["rays_shadow"], cnt["nodes_shadow"], cnt["tris_shadow"], cnt["n_shadow_launches"]),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        ms, launches, rays, nodes, tris, cnt_launches = kernels[dom]
        node_bytes = timed["bvh_node_bytes"] / max(timed["bvh_nodes"], 1)  # 80 B wide nodes (64 B for the binary fallback)
        # algorithmic bytes of one step (counted replay) x timed steps / timed launches = bytes per launch; launch time = HIP-event
        # total / timed launches.  Both sides use the timed launch count, so achieved = bytes per step x steps / kernel seconds.
        bytes_per_step = rays * (RAY_BYTES + HIT_BYTES) + nodes * node_bytes + tris * TRI_BYTES
        bytes_per_launch = bytes_per_step * args.steps / max(launches, 1)
        avg_ms = ms / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc run (counters cannot be read from inside this
        # process); the committed summary of the latest such run is quoted when it is for the same dominant kernel
        traffic = None
        try:
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if tf:
                tj = json.load(open(tf[-1]))
                if tj.get("kernel", "").startswith(dom):
                    traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "Msamples/s at 1920x1080, max_depth=8", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: 1M random-triangle soup (PCG32 seed 0x853c49e6748fea9b), Hosek sky turbidity 3 albedo 0.3, 1920x1080, max_depth 8, seed 1",
                       "spp_per_step": args.spp, "triangles": N_TRIS, "parallelism": f"pixel-tile x{world}" if world > 1 else "single GPU",
                       "gather": "RCCL all_gather of packed float4 beauty tiles, inside the timed region" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dom + "_stream", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "launches": int(launches), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "per_ray": {"nodes": round(nodes / max(rays, 1), 2), "triangles": round(tris / max(rays, 1), 2), "bytes": round(bytes_per_step / max(rays, 1), 1)},
                         "note": "rank 0 shard" if world > 1 else "whole frame"},
            "kernel_ms_per_step": {"trace_closest": round(timed["trace_closest_ms"] / args.steps, 3), "trace_secondary": round(timed["trace_shadow_ms"] / args.steps, 3),
                                   "shade": round(timed["shade_ms"] / args.steps, 3), "tail": round(timed["tail_ms"] / args.steps, 3), "render_total": round(timed["render_ms"] / args.steps, 3)},
            "bvh": {"build_ms": round(timed["bvh_build_ms"], 2), "nodes": timed["bvh_nodes"], "node_bytes": timed["bvh_node_bytes"], "tri_bytes": timed["bvh_tri_bytes"], "depth": timed["bvh_depth"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
