#!/usr/bin/env python3
"""rccl_gather_probe.py -- the collective leg of bench.py --gpus N on its own: what the driver's multi-GPU run would hit first, failing loudly.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_gather_probe.py [--width 1920 --height 1080] [--backend nccl|gloo]

Every rank: initialises the process group exactly as bench.py does (RCCL with device_id = LOCAL_RANK), runs fredholm_amd.distributed.preflight (environment,
device binding, equal shard shapes, one checked gather of the packed-shard size), then times 20 gathers of the packed float4 beauty shard of a width x height
frame split into interleaved 32x32 tiles.  Rank 0 prints one JSON line.  Needs no renderer and no scene: a failure here is a failure of the launch, not of the path tracer.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--backend", default=os.environ.get("FH_BENCH_BACKEND", "nccl"))
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from fredholm_amd import distributed as D

    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        if k not in os.environ:
            raise SystemExit(f"rccl_gather_probe: {k} is not set -- start me with python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P ...")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    if args.backend == "nccl":
        if not torch.cuda.is_available():
            raise SystemExit("rccl_gather_probe: no GPU visible (backend nccl = RCCL needs one per rank)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rccl_gather_probe: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group(args.backend)
    pad = D.max_owned(args.width, args.height, world)
    first = D.preflight(dist, device, pad)
    dev = device if args.backend == "nccl" else torch.device("cpu")
    packed = torch.full((pad, 4), float(rank), dtype=torch.float32, device=dev)
    outs = [torch.zeros_like(packed) for _ in range(world)] if rank == 0 else None
    ts = []
    for _ in range(20):
        dist.barrier()
        t0 = time.perf_counter()
        dist.gather(packed, outs, dst=0)
        if args.backend == "nccl":
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    if rank == 0:
        ok = all(bool((outs[k] == float(k)).all().item()) for k in range(world))
        print(json.dumps({"probe": "gather of packed float4 beauty shards to rank 0", "backend": args.backend, "world": world, "frame": [args.width, args.height], "bytes_per_rank": pad * 16,
                          "first_gather_ms": round(first * 1e3, 3), "median_ms": round(ts[len(ts) // 2] * 1e3, 3), "min_ms": round(ts[0] * 1e3, 3), "data_ok": ok}), flush=True)
        if not ok:
            raise SystemExit("rccl_gather_probe: gathered data is wrong")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
