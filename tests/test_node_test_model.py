"""The 8-wide node test of the traversal kernels (fredholm_amd/csrc/fh_trace.h: node8_test) as a numpy model, checked on the CPU against the exact slab test.

The HIP code itself is checked on the GPU (tools/micro/issue_peak.hip compares it with the slab test in double precision on 16.7 M random pairs, and every
-m gpu parity test traverses with it); this file pins the ALGORITHM -- distances in units of the ray's limit, clamped to [0, 1], a strict comparison of
max3(near) and min3(far), near planes moved in and far planes out by 2^-21 of the axis' offset -- and the properties the kernels rely on:
it never misses a child the ray enters before tmax, it flags no proper box for a negative tmax, an empty slot (lo 255, hi 0) is never flagged for a limit >= 0.
The model computes in float32 with the fused multiply-add taken in float64 and rounded once (exact products, one rounding)."""
import numpy as np
import pytest

f32 = np.float32
SLACK = f32(4.76837158203125e-7)  # 2^-21
DOWN = f32(0.99999976158142090)   # 1 - 2^-22


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def clamp01(x):
    return np.where(np.isnan(x), f32(0), np.clip(x, f32(0), f32(1))).astype(f32)


def safe_reciprocal(d):
    d = np.where(np.abs(d) < f32(1e-20), np.copysign(f32(1e-20), d), d).astype(f32)
    return (f32(1) / d).astype(f32)


def node8_test(o, inv, origin_words, lo, hi, tmax):
    """o, inv: [n, 3]; origin_words: [n, 3] uint32 (origin float whose low mantissa byte is the scale exponent); lo, hi: [n, 3, 8] uint8; tmax: [n].
    Returns [n, 8] bool."""
    with np.errstate(all="ignore"):
        tl = np.maximum(tmax.astype(f32).view(np.uint32), f32(1e-12).view(np.uint32)).view(f32)  # unsigned maximum of the bit patterns: a floor for tmax >= +0 only
        tl = np.fmin(tl, f32(1e30)).astype(f32)
        rt = ((f32(1) / tl).astype(f32) * DOWN).astype(f32)
        i = (inv * rt[:, None]).astype(f32)
        p = origin_words.view(f32)
        k = np.abs((origin_words << np.uint32(23)).view(f32))  # 2^(e - 127); bit 8 of the word lands in the sign
        s = (k * i).astype(f32)
        off = ((p - o).astype(f32) * i).astype(f32)
        far_off = fma(np.abs(off), np.broadcast_to(SLACK, off.shape), off)
        near_off = fma(np.abs(off), np.broadcast_to(-SLACK, off.shape), off)
        neg = inv < 0
        near_q = np.where(neg[:, :, None], hi, lo).astype(f32)
        far_q = np.where(neg[:, :, None], lo, hi).astype(f32)
        tn = clamp01(fma(near_q, np.broadcast_to(s[:, :, None], near_q.shape), np.broadcast_to(near_off[:, :, None], near_q.shape)))
        tf = clamp01(fma(far_q, np.broadcast_to(s[:, :, None], far_q.shape), np.broadcast_to(far_off[:, :, None], far_q.shape)))
        d = (tn.max(axis=1) - tf.min(axis=1)).astype(f32)
        return np.signbit(d)


def exact_slab(o, inv, origin_words, lo, hi, tmax, margin):
    """children the ray certainly enters in [0, tmax]: exact arithmetic on the dequantised boxes (float64), with a relative margin between entry and exit"""
    p = origin_words.view(f32).astype(np.float64)
    k = np.ldexp(1.0, (origin_words & np.uint32(0xFF)).astype(np.int64) - 127)
    blo = p[:, :, None] + k[:, :, None] * lo.astype(np.float64)
    bhi = p[:, :, None] + k[:, :, None] * hi.astype(np.float64)
    a = (blo - o.astype(np.float64)[:, :, None]) * inv.astype(np.float64)[:, :, None]
    b = (bhi - o.astype(np.float64)[:, :, None]) * inv.astype(np.float64)[:, :, None]
    tn = np.maximum(np.minimum(a, b).max(axis=1), 0.0)
    tf = np.minimum(np.maximum(a, b).min(axis=1), tmax.astype(np.float64)[:, None])
    empty = (lo > hi).any(axis=1)
    return (~empty) & (tn * (1.0 + margin) + 1e-30 < tf * (1.0 - margin))


def random_cases(rng, n, far=False):
    o = (rng.random((n, 3), dtype=f32) * f32(4) - f32(2)).astype(f32)
    if far:
        o = (o * f32(1.0e4)).astype(f32)
    d = (rng.random((n, 3), dtype=f32) * f32(2) - f32(1)).astype(f32)
    d[::7, 0] = 0.0
    d[::11, 1] = -0.0
    e = rng.integers(100, 130, size=(n, 3)).astype(np.uint32)
    words = ((rng.random((n, 3), dtype=f32) * f32(2) - f32(1)).astype(f32).view(np.uint32) & np.uint32(0xFFFFFF00)) | e
    if far:  # from 10^4 scene sizes away a random direction never meets the node: aim at a point inside it
        target = words.view(f32).astype(np.float64) + np.ldexp(255.0, e.astype(np.int64) - 127) * rng.random((n, 3))
        aim = target - o.astype(np.float64)
        d = (aim / np.linalg.norm(aim, axis=1, keepdims=True)).astype(f32)
    inv = safe_reciprocal(d)
    lo = rng.integers(0, 256, size=(n, 3, 8)).astype(np.uint8)
    hi = rng.integers(0, 256, size=(n, 3, 8)).astype(np.uint8)
    lo[::5], hi[::5] = 0, 255  # every child the whole node box
    return o, inv, words, lo, hi


@pytest.mark.parametrize("far", [False, True])
def test_no_child_the_ray_enters_is_missed(far):
    rng = np.random.default_rng(20260904 + int(far))
    n = 200_000
    o, inv, words, lo, hi = random_cases(rng, n, far)
    tmax = np.where(np.arange(n) % 2 == 0, f32(1e9), (rng.random(n, dtype=f32) * f32(3)).astype(f32)).astype(f32)
    tmax[::9] = f32(3.0e38)  # a ray without a limit
    got = node8_test(o, inv, words, lo, hi, tmax)
    must = exact_slab(o, inv, words, lo, hi, tmax, 1e-5)
    assert not (must & ~got).any()
    assert must.sum() > n // 50  # the cases do exercise hits
    # conservative, not useless: whatever it flags beyond the certain hits is a box the ray grazes (entry within 0.2 % of exit: the random boxes include
    # slabs a few 2^-27 thick, which the margin of `must` leaves out)
    # (rays from 10^4 scene sizes away are given 2^-21 of that distance as slack on purpose: their own rounding is of that order)
    if not far:
        grazed = exact_slab(o, inv, words, lo, hi, tmax, -1e-3)
        assert (got & ~grazed).sum() <= 0.002 * max(int(got.sum()), 1)


def test_negative_limit_flags_nothing_and_empty_slots_are_never_flagged():
    rng = np.random.default_rng(7)
    n = 100_000
    o, inv, words, lo, hi = random_cases(rng, n)
    # a negative limit reverses every scaled interval: no box with lo <= hi on some axis can be flagged.  (A box inverted on all three axes can -- in a tree
    # that is an empty slot, whose triangle slot holds the degenerate triangle no ray hits.)
    flagged = node8_test(o, inv, words, lo, hi, np.full(n, -1.0, f32))
    assert not (flagged & (lo <= hi).any(axis=1)).any()
    lo[:, :, 3], hi[:, :, 3] = 255, 0  # slot 3 empty everywhere (the builder's inverted box)
    got = node8_test(o, inv, words, lo, hi, np.full(n, 1e9, f32))
    assert not got[:, 3].any()


def test_a_best_hit_at_the_origin_still_enters_the_boxes_that_hold_the_origin():
    """tmax = +0 is a ray whose best hit so far is at its origin (an origin on a triangle).  A second triangle there with a lower face id must still be found, so the boxes
    that hold the origin must be flagged (the unit has a floor of 1e-12), while boxes that start later are not, a negative limit still flags nothing and a NaN limit still
    counts as none."""
    rng = np.random.default_rng(11)
    n = 50_000
    o, inv, words, lo, hi = random_cases(rng, n)
    zero = np.zeros(n, f32)
    got = node8_test(o, inv, words, lo, hi, zero)
    p = words.view(f32).astype(np.float64)
    k = np.ldexp(1.0, (words & np.uint32(0xFF)).astype(np.int64) - 127)
    blo, bhi = p[:, :, None] + k[:, :, None] * lo.astype(np.float64), p[:, :, None] + k[:, :, None] * hi.astype(np.float64)
    oo = o.astype(np.float64)[:, :, None]
    holds = ((blo < oo - 1e-6) & (oo + 1e-6 < bhi)).all(axis=1)   # the origin strictly inside the box
    assert holds.sum() > 200 and not (holds & ~got).any()
    a, b = (blo - oo) * inv.astype(np.float64)[:, :, None], (bhi - oo) * inv.astype(np.float64)[:, :, None]
    later = np.minimum(a, b).max(axis=1) > 1e-6                     # the ray enters the box after t = 1e-6: not a candidate for a hit at t = 0
    assert not (got & later).any()
    assert not (node8_test(o, inv, words, lo, hi, np.full(n, -0.5, f32)) & (lo <= hi).any(axis=1)).any()
    nan = node8_test(o, inv, words, lo, hi, np.full(n, np.nan, f32))
    assert not (exact_slab(o, inv, words, lo, hi, np.full(n, 1e30, f32), 1e-5) & ~nan).any()
