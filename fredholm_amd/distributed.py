"""Pixel-tile sharding of one frame over the GPUs of a node, one process per GPU.

The reference is single-GPU (one optixLaunch over W x H, renderer.h:730-733).  Pixels never
communicate during rendering and every sampler key is a function of (pixel, sample, slot, seed)
(pt.cu:378-399), so any pixel -> rank mapping gives bit-identical per-pixel results: each rank
renders the tiles t with t % world == rank (interleaved for load balance), keeps its own
sample counters, and the only exchange is one gather of the packed owned pixels per presented
frame (RCCL all_gather over xGMI; `gloo` in the CPU tests).
"""
import numpy as np


def tile_ownership(width, height, rank, world, tile_w=32, tile_h=32):
    """Image indices (x + width*y) owned by `rank`, in the order the library packs them.

    Must stay identical to rebuild_ownership() in fredholm_amd/csrc/capi.hip (checked on the GPU
    by tests/test_gpu_parity.py::test_tile_ownership_matches_library)."""
    tx = (width + tile_w - 1) // tile_w
    ty = (height + tile_h - 1) // tile_h
    out = []
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        xs = np.arange(x0, min(x0 + tile_w, width), dtype=np.uint32)
        ys = np.arange(y0, min(y0 + tile_h, height), dtype=np.uint32)
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
