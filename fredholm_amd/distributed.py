"""Pixel-tile sharding of one frame over the GPUs of a node, one process per GPU.

The reference is single-GPU (one optixLaunch over W x H, renderer.h:730-733).  Pixels never
communicate during rendering and every sampler key is a function of (pixel, sample, slot, seed)
(pt.cu:378-399), so any pixel -> rank mapping gives bit-identical per-pixel results: each rank
renders the tiles t with t % world == rank (interleaved for load balance), keeps its own
sample counters, and the only exchange is one gather of the packed owned pixels per presented
frame (RCCL all_gather over xGMI; `gloo` in the CPU tests).
"""
import numpy as np


PIXEL_BLOCK = 8  # capi.hip: kPixelBlock


def tile_ownership(width, height, rank, world, tile_w=32, tile_h=32, block=PIXEL_BLOCK):
    """Image indices (x + width*y) owned by `rank`, in the order the library packs them: tiles t % world == rank in row-major tile order; inside a tile
    `block` x `block` pixel blocks, row-major, and row-major inside a block (a wave of the camera-ray kernel then holds a compact patch of the image).

    Must stay identical to owned_list() in fredholm_amd/csrc/capi.hip (checked on the GPU
    by tests/test_gpu_parity.py::test_tile_ownership_matches_library_and_shards_reassemble)."""
    tx = (width + tile_w - 1) // tile_w
    ty = (height + tile_h - 1) // tile_h
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        x1, y1 = min(x0 + tile_w, width), min(y0 + tile_h, height)
        for by in range(y0, y1, min(block, tile_h)):
            for bx in range(x0, x1, min(block, tile_w)):
                xs = np.arange(bx, min(bx + block, x1), dtype=np.uint32)
                ys = np.arange(by, min(by + block, y1), dtype=np.uint32)
                out.append((xs[None, :] + np.uint32(width) * ys[:, None]).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint32)


def max_owned(width, height, world, tile_w=32, tile_h=32):
    return max(tile_ownership(width, height, r, world, tile_w, tile_h).size for r in range(world))


def gather_packed(local_packed, pad_to, dist, group=None):
    """all_gather equally padded packed shards; returns the list of per-rank tensors (every rank)."""
    import torch

    world = dist.get_world_size(group)
    fpp = local_packed.shape[1]
    padded = torch.zeros((pad_to, fpp), dtype=local_packed.dtype, device=local_packed.device)
    padded[: local_packed.shape[0]] = local_packed
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded, group=group)
    return outs


def assemble(width, height, shards, tile_w=32, tile_h=32):
    """Scatter gathered shards (list of [pad_to, fpp] tensors/arrays) into a full [H, W, fpp] image (host side)."""
    world = len(shards)
    first = np.asarray(shards[0])
    img = np.zeros((height * width, first.shape[1]), dtype=first.dtype)
    for r in range(world):
        own = tile_ownership(width, height, r, world, tile_w, tile_h)
        img[own] = np.asarray(shards[r])[: own.size]
    return img.reshape(height, width, -1)


def preflight(dist, device, packed_rows, fpp=4):
    """Fail loudly, before any frame is rendered, on what would otherwise hang or corrupt the RCCL gather of bench.py: a rendezvous the environment
    does not describe, a process group bound to another device than the renderer's, ranks that disagree on the shard size, a collective that does
    not move the bytes it should.  `dist` is an initialised torch.distributed, `device` this rank's torch device, `packed_rows` the rows of the
    padded packed shard.  Returns the seconds one gather of that size took."""
    import os
    import time

    import torch

    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in os.environ:
            raise RuntimeError(f"distributed launch without {k} in the environment (start with python -m torch.distributed.run --master-addr 127.0.0.1 --master-port P ...)")
    rank, world = dist.get_rank(), dist.get_world_size()
    if world != int(os.environ["WORLD_SIZE"]) or rank != int(os.environ["RANK"]):
        raise RuntimeError(f"process group says rank {rank} of {world}, environment says {os.environ['RANK']} of {os.environ['WORLD_SIZE']}")
    on_gpu = dist.get_backend() == "nccl"
    if on_gpu:
        if device.type != "cuda" or torch.cuda.current_device() != device.index:
            raise RuntimeError(f"rank {rank}: RCCL needs the current device to be the renderer's ({device}), it is cuda:{torch.cuda.current_device()}")
        if torch.cuda.device_count() < world and os.environ.get("FH_ALLOW_SHARED_GPU") != "1":
            raise RuntimeError(f"{world} RCCL ranks on {torch.cuda.device_count()} visible GPU(s): one process per GPU (FH_BENCH_BACKEND=gloo shares a GPU for functional tests)")
    dev = device if on_gpu else torch.device("cpu")
    # every rank must bring the same shard shape: a gather of unequal tensors hangs or truncates
    mine = torch.tensor([int(packed_rows), int(fpp)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    got = [tuple(int(v) for v in t.tolist()) for t in sizes]
    if len(set(got)) != 1:
        raise RuntimeError(f"packed shard shapes differ across ranks: {got} (every rank pads to distributed.max_owned)")
    # one gather of the real size with a pattern only the right rank can have produced
    payload = torch.full((int(packed_rows), int(fpp)), float(rank + 1), dtype=torch.float32, device=dev)
    outs = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    if on_gpu:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.gather(payload, outs, dst=0)
    if on_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        for k in range(world):
            if not bool((outs[k] == float(k + 1)).all().item()):
                raise RuntimeError(f"gather returned wrong data for rank {k}")
    dist.barrier()
    return dt
