#!/usr/bin/env python3
"""Generate tests/golden/ref_lut_math_post.npz by RUNNING the reference's own functions, built for the host from its sources by
oracle/Makefile into oracle/_ref/libref_lut_math_post.so:
  fredholm/modules/lut.cu:957-1081   compute_directional_albedo_reflection / _reflection_ior1 / _sheen (+ the three raw tables, :5-955)
  fredholm/modules/math.cu:7-35,90-118   orthonormal_basis, world_to_local, local_to_world, rgb_to_luminance, cartesian_to_spherical
  fredholm/kernels/include/kernels/post-process.h:13-124   rgb_to_luminance, uchimura, linear_to_srgb, compute_EV100,
                                                           convert_EV100_to_exposure, and the tail of tone_mapping_kernel
Needs /root/reference (at build time of the library).  The fixture is data: float32 inputs and the reference's float32 outputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def inputs():
    rng = np.random.default_rng(20261004)
    f32 = np.float32
    g = np.linspace(0.0, 1.0, 33, dtype=f32)
    edge = np.array([-0.25, 0.0, 1e-7, 0.03125, 0.0625 - 1e-7, 0.0625, 0.5, 0.9375, 1.0 - 1e-7, 1.0, 1.25], dtype=f32)
    uu, vv, ff = np.meshgrid(g, g, np.array([0.0, 0.04, 0.25, 0.8, 1.0], dtype=f32), indexing="ij")
    refl = np.stack([uu.ravel(), vv.ravel(), ff.ravel()], 1)
    eu, ev = np.meshgrid(edge, edge, indexing="ij")
    refl = np.concatenate([refl, np.stack([eu.ravel(), ev.ravel(), np.full(eu.size, 0.04, f32)], 1), np.stack([-eu.ravel(), ev.ravel(), np.full(eu.size, 0.5, f32)], 1)])
    su, sv = np.meshgrid(g, g, indexing="ij")
    sheen = np.concatenate([np.stack([su.ravel(), sv.ravel()], 1), np.stack([eu.ravel(), ev.ravel()], 1), np.stack([-eu.ravel(), ev.ravel()], 1)])
    dirs = unit(rng.normal(size=(3000, 3)))
    axes = np.array([[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 1, -0.0], [1e-4, 1e-4, -1], [0.6, 0.8, 0.0], [0.6, 0.8, -0.0]], dtype=f32)
    dirs = np.concatenate([unit(axes), dirs]).astype(f32)
    t = unit(rng.normal(size=(3000, 3)))
    frames = np.concatenate([rng.normal(size=(3000, 3)).astype(f32), t, dirs[:3000], unit(rng.normal(size=(3000, 3)))], 1).astype(f32)
    rgb = np.concatenate([np.exp(rng.uniform(-14, 5, (2000, 3))), np.array([[0, 0, 0], [0.22, 0.22, 0.22], [0.532, 0.532, 0.532], [1, 1, 1], [0.0031308, 0.0031307, 0.0031309]])]).astype(f32)
    sweep = np.exp(np.linspace(-16, 6, 1024)).astype(f32)
    rgb = np.concatenate([rgb, np.stack([sweep, sweep[::-1], np.roll(sweep, 7)], 1)]).astype(f32)
    expo = np.stack([rng.uniform(1, 22, 1000), np.exp(rng.uniform(-9, 1, 1000)), rng.choice([50, 80, 100, 200, 400, 800, 1600, 3200, 6400], 1000)], 1).astype(f32)
    expo = np.concatenate([expo, np.array([[1, 1, 80], [1, 1, 100], [1, 1, 400], [16, 0.008, 100]], dtype=f32)])
    tail = np.concatenate([np.concatenate([rgb, np.full((rgb.shape[0], 1), 80.0, f32)], 1), np.concatenate([rgb[:1000], np.full((1000, 1), 400.0, f32)], 1)]).astype(f32)
    return {"albedo_reflection": refl.astype(f32), "albedo_sheen": sheen.astype(f32), "onb": dirs, "to_local": frames, "to_world": frames, "spherical": dirs,
            "luminance": rgb, "post_luminance": rgb, "uchimura": rgb, "linear_to_srgb": rgb, "exposure": expo, "tone_map_tail": tail}


def main():
    assert O.ref_lut_math_post() is not None, "oracle/_ref/libref_lut_math_post.so is missing: run `make -C oracle` where /root/reference exists"
    data = {}
    for kind, x in inputs().items():
        data["in_" + kind] = x
        data["out_" + kind] = O.ref_math(kind, x)
    refl, sheen = O.ref_lut_tables()
    data["table_reflection"], data["table_sheen"] = refl, sheen
    # REFLECTION_IOR1_LUT (lut.cu:94-916), 16^3: the reference's own trilinear fetcher (lut.cu:1006-1045) returns the raw entry [i + 16 j + 256 k] at
    # (w.y, roughness, eta) = (i, j, k) / 16 (all three fractions are exactly 0 there); plus the fetcher itself at random points
    g = (np.arange(16) / 16.0).astype(np.float32)
    kk, jj, ii = np.meshgrid(g, g, g, indexing="ij")
    data["table_reflection_ior1"] = O.ref_albedo_reflection_ior1(np.stack([ii.ravel(), jj.ravel(), kk.ravel()], 1))
    rng = np.random.default_rng(20261005)
    x = rng.uniform(-0.1, 1.1, (2000, 3)).astype(np.float32)
    data["in_albedo_reflection_ior1"], data["out_albedo_reflection_ior1"] = x, O.ref_albedo_reflection_ior1(x)
    out = os.path.join(ROOT, "tests", "golden", "ref_lut_math_post.npz")
    np.savez_compressed(out, **data)
    print({k: v.shape for k, v in data.items()}, "->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
