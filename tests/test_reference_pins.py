"""Reference pins: the checker (CPU) and the HIP library (GPU) against outputs of the REFERENCE's own code.

tests/golden/ref_lut_math_post.npz holds inputs and the outputs of the reference's lut.cu, math.cu and kernels/post-process.h,
compiled for the host from /root/reference by oracle/Makefile into oracle/_ref/libref_lut_math_post.so and run by
tests/golden/gen_ref_golden.py.  Functions made of +, -, *, / only (LUT fetches, shading-frame helpers, luminance) must agree with the
reference BIT FOR BIT; functions that call libm (acosf / atan2f / powf / expf / log2f: glibc on the reference side, include/fh_elementary.h on
ours) within the stated ulp bound.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_lut_math_post.npz")
EXACT = ("albedo_reflection", "albedo_sheen", "onb", "to_local", "to_world", "luminance", "post_luminance")
# kind -> max ulp distance from the reference's float32 result (glibc libm vs fh_elementary, each <= 2 ulp, chained)
ULP = {"spherical": 4, "uchimura": 4, "linear_to_srgb": 4, "exposure": 16, "tone_map_tail": 6}


def ulp_distance(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def check(kind, got, want):
    assert got.shape == want.shape
    if kind in EXACT:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{kind}: not bit-identical to the reference"
    else:
        d = ulp_distance(got, want)
        # near a zero crossing a fixed absolute error is many ulps: phi ~ 0, or a tone-mapped value that underflows
        small = np.abs(want) < 1e-6
        assert d[~small].max() <= ULP[kind], f"{kind}: {d[~small].max()} ulp from the reference"
        assert np.abs(got - want)[small].max(initial=0.0) <= 1e-9


@pytest.mark.parametrize("kind", list(EXACT) + list(ULP))
def test_checker_matches_reference_built_functions(oracle, golden, kind):
    check(kind, oracle.math(kind, golden["in_" + kind]), golden["out_" + kind])


def test_albedo_tables_are_the_reference_tables(golden):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fredholm_amd", "data")
    assert np.array_equal(np.fromfile(os.path.join(d, "lut_reflection.f32"), np.float32), golden["table_reflection"])  # lut.cu:5-93
    assert np.array_equal(np.fromfile(os.path.join(d, "lut_sheen.f32"), np.float32), golden["table_sheen"])            # lut.cu:917-955


def test_fixture_is_what_the_reference_build_produces(oracle, golden):
    """where oracle/_ref was built (this container), re-running the reference reproduces the committed fixture bit for bit"""
    if oracle.ref_lut_math_post() is None:
        pytest.skip("oracle/_ref/libref_lut_math_post.so not built (no /root/reference at build time)")
    for kind in list(EXACT) + list(ULP):
        again = oracle.ref_math(kind, golden["in_" + kind])
        assert np.array_equal(again.view(np.uint32), golden["out_" + kind].view(np.uint32)), kind


def test_post_process_image_follows_the_pinned_helpers(oracle):
    """the checker's whole post chain on an image = the reference's pixel addressing + the pinned per-pixel tail: with bloom off and no
    aberration every covered pixel equals tone_map_tail of the pixel tone_mapping_kernel fetches -- which, computed in float as
    uv.x * width + width * (uv.y * height) (post-process.cu:133-135), is not always the pixel itself"""
    rng = np.random.default_rng(5)
    h, w = 40, 56
    img = np.exp(rng.uniform(-6, 3, (h, w, 4))).astype(np.float32)
    out = oracle.post_process(img, False, 2.0, 5.0, 80.0, 0.0)
    f = np.float32
    jj, ii = np.meshgrid(np.arange(32), np.arange(48), indexing="ij")
    uvx, uvy = ii.astype(f) / f(w), jj.astype(f) / f(h)
    idx = (uvx * f(w) + f(w) * (uvy * f(h))).astype(f).astype(np.int64)
    assert (idx != ii + w * jj).any()  # the quirk is exercised
    src = img.reshape(-1, 4)[idx.ravel(), :3]
    tail = oracle.math("tone_map_tail", np.concatenate([src, np.full((src.shape[0], 1), 80.0, np.float32)], 1)).reshape(32, 48, 3)
    assert np.array_equal(out[:32, :48, :3], tail)
    assert (out[32:] == 0).all() and (out[:, 48:] == 0).all()  # floor-division grid, post-process.cu:9-11


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kind", list(EXACT) + list(ULP))
def test_hip_matches_reference_built_functions(renderer, oracle, golden, kind):
    from fredholm_amd import native as N
    k, si, so = oracle.MATH_KINDS[kind]  # kinds and widths of fh_kat_math (include/fredholm_hip.h)
    x = np.ascontiguousarray(golden["in_" + kind], dtype=np.float32).reshape(-1, si)
    got = np.zeros((x.shape[0], so), np.float32)
    N.check(renderer._ctx, N.lib().fh_kat_math(renderer._ctx, k, int(x.shape[0]), N.ptr(x), N.ptr(got)), "fh_kat_math")
    check(kind, got, golden["out_" + kind])
    # and the HIP path equals the checker bit for bit (same elementary functions on both sides)
    assert np.array_equal(got.view(np.uint32), oracle.math(kind, golden["in_" + kind]).view(np.uint32))
