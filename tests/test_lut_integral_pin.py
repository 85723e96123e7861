"""A pin of the lobes to reference DATA: the reference's albedo tables are integrals of its own lobes.

`fredholm/modules/lut.cu:5-93` (REFLECTION_LUT, 16 x 16 x 2), `:917-955` (SHEEN_LUT, 16 x 16) and `:94-916` (REFLECTION_IOR1_LUT, 16^3, declared but never
fetched on the live path) are the numbers in the reference tree that were produced BY the GGX reflection lobe (`bxdf.cu:428-518`) and the sheen lobe
(`bxdf.cu:743-822`): directional albedos over (cos theta_o, roughness[, eta]).  The baking code is not in the tree; what the tables hold was found by trying:

  * every entry is the Monte-Carlo mean of  f(wo, wi) |cos theta_i| / pdf  over the lobe's OWN `sample()` -- half vector from the visible-normal /
    cosine distribution, wi = reflect(wo, wh) -- WITHOUT rejecting reflected directions below the horizon (the lobes use |cos|, so such samples
    contribute), at the CELL CENTRES ((i + 0.5) / 16, (j + 0.5) / 16);
  * REFLECTION_LUT.x has Fresnel = 1, .y Schlick's (1 - c)^5 (the fetch returns F0 x + (1 - F0) y).

So a replay of that estimator through `BSDF::sample` / `eval` / `eval_pdf` with a material that isolates one lobe must land on the table: a misread
alpha mapping, a missing 1/4, a wrong Lambda or L(x), a wrong D normalisation or a wrong pdf moves the result by far more than the tolerances below.
The same replay runs through the CPU checker (here, `-m "not gpu"`) and through the HIP library's `fh_kat_bsdf` (`-m gpu`).

Observed (and asserted with a little room):
  sheen       the 225 cells with cos >= 0.09 and roughness >= 0.09: mean +0.01 %, rms 0.16 %, worst +0.9 %; roughness cell 0 (1/32: a spike, the replay converges from
              below with the number of draws) within 5 %; cos cell 0 (cos 0.031, grazing) -10 ... +33 %: the table's estimate there is a handful of samples
  reflection  .x against the conductor lobe at reflectivity 0.99 (the BSDF clamps base colour to 0.99, bsdf.cu:97): -1.0 ... -2.2 % on the 192 cells with cos >= 0.28,
              i.e. the 0.99 itself and nothing else; down to -9 % at cos 0.031
  reflection  0.04 x + 0.96 y against the dielectric lobe at ior 1.5: -6 % ... +29 % -- the table's Schlick Fresnel against the lobe's exact one
              (exact F(58 deg) = 0.083, Schlick 0.063), +-5 % at normal incidence where Schlick is exact; stated, not tuned away
  reflection  the dielectric lobe with its own Fresnel divided out (round 6: what is left is D, G2 / G1 and the visible-normal sampler): .x within -0.6 ... +0.2 % on ALL 256 cells
              (rms 0.2 %); .y as that times Schlick's (1 - c)^5: within 0.005 absolute, 0.8 % on the 77 entries >= 0.05
  reflection, eta < 1 (round 6)  REFLECTION_IOR1_LUT, lut.cu:94-916: 16^3 over (cos theta_o, roughness, eta), declared and never fetched by the reference's live
              path (lut.cu:1038-1045), is the SAME estimator applied to `MicrofacetReflectionDielectric(ior = eta, roughness)` with its EXACT Fresnel, total internal
              reflection included (bxdf.cu:274-283: F = 1 where eta^2 + c^2 < 1) -- its eta-cell-0 slice equals REFLECTION_LUT.x to four digits.  The constructor of
              `BSDF` only ever passes 1.5 or 1 / 1.5 (bsdf.cu:16-18), so the replay goes through `orc_bsdf_ior` / `fh_kat_bsdf_ior`, which set the relative index the
              lobe classes take as an argument.  All 4096 cells: |replay - table| <= 0.0085 (rms 0.0018) with 48 x 48 draws; on the 3759 cells whose table value is
              >= 0.05: rms 0.35 %, 15 cells beyond 2 %, worst 4.8 %, all on the edge of total internal reflection where a cell is the mean of a step.  The cells far
              BELOW 0.05 (eta -> 1, smooth: F0 ~ 2e-4) are dominated by the few microfacets in GGX's tail that reflect totally; the replay converges to the table
              from below with the number of draws (-43 % at 48^2, -4 % at 384^2 on the cell checked here), so they are held to the absolute bound only.
and the hemispherical integral proper (directions below the horizon rejected) is what the table holds only for smooth lobes: at roughness 1 and
normal incidence the table is TWICE the integral (0.626 against 1 - ln 2 = 0.307): the reference's albedo tables overestimate rough lobes, by construction.
"""
import ctypes as C
import os

import numpy as np
import pytest

from fredholm_amd.native import default_materials

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "ref_lut_math_post.npz"))
T_REFL = GOLD["table_reflection"].reshape(16, 16, 2)  # [roughness j][cos i][x, y]  (lut.cu:957-994: idx = 2 i + 32 j)
T_SHEEN = GOLD["table_sheen"].reshape(16, 16)


def _material(**kw):
    m = default_materials(1)
    for k, v in kw.items():
        m[k] = v
    return m


# materials under which BSDF::eval / sample / eval_pdf are ONE lobe with weight 1 (bsdf.cu:129-212: the other terms are multiplied by 0)
def metal(r):
    return _material(metalness=1.0, base_color=(1, 1, 1), specular_color=(1, 1, 1), specular_roughness=r, diffuse=0.0)


def dielectric(r):
    return _material(specular=1.0, specular_color=(1, 1, 1), specular_roughness=r, diffuse=0.0)


def sheen(r):
    return _material(specular=0.0, diffuse=0.0, sheen=1.0, sheen_color=(1, 1, 1), sheen_roughness=r)


def _strata(k):
    a = (np.arange(k, dtype=np.float64) + 0.5) / k
    return np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2).astype(np.float32)


def replay(bsdf, material, cosines, k=64, reject_below_horizon=False):
    """the baker's estimator for every cos theta_o of `cosines`: mean of f |cos theta_i| / pdf over k x k stratified draws of the lobe's own sampler"""
    U = _strata(k)
    n = U.shape[0]
    wo = np.concatenate([np.tile(np.array([np.sqrt(max(1.0 - c * c, 0.0)), c, 0.0], np.float32), (n, 1)) for c in cosines])
    out = bsdf(material, wo, wo, np.full(wo.shape[0], 0.5, np.float32), np.tile(U, (len(cosines), 1)))
    wi, f, pdf = out[:, 4:7], out[:, 7].astype(np.float64), out[:, 10].astype(np.float64)
    ok = (pdf > 0) & np.isfinite(pdf) & np.isfinite(f)
    if reject_below_horizon:
        ok &= wi[:, 1] > 0
    w = np.where(ok, f * np.abs(wi[:, 1].astype(np.float64)) / np.where(pdf > 0, pdf, 1.0), 0.0)
    return w.reshape(len(cosines), n).mean(axis=1)


CENTRES = (np.arange(16) + 0.5) / 16.0


def table_errors(bsdf, lobe, table, k=64):
    """relative deviation of the replay from the table at the cell centres: array [roughness cell][cos cell]"""
    return np.array([replay(bsdf, lobe(float(CENTRES[j])), CENTRES, k) / table[j] - 1.0 for j in range(16)])


def check_tables(bsdf):
    # ---- sheen: 128 x 128 draws per cell (the lobe at roughness 1/32 is a spike at grazing half vectors: 64 x 64 draws are 10 % short there)
    e = table_errors(bsdf, sheen, T_SHEEN, 128)
    inner = e[1:, 1:]
    worst = np.unravel_index(np.abs(inner).argmax(), inner.shape)
    print(f"sheen, 225 cells with cos >= 0.09 and roughness >= 0.09: mean {inner.mean():+.5f} rms {np.sqrt((inner ** 2).mean()):.5f} worst {inner[worst]:+.4f} at roughness cell {worst[0] + 1}, cos cell {worst[1] + 1}; "
          f"roughness cell 0: {e[0, 1:].min():+.3f} ... {e[0, 1:].max():+.3f}; cos cell 0 (cos 0.031): {e[:, 0].min():+.3f} ... {e[:, 0].max():+.3f}")
    assert abs(inner.mean()) < 0.002 and np.sqrt((inner ** 2).mean()) < 0.004 and np.abs(inner).max() < 0.02
    assert np.abs(e[0, 1:]).max() < 0.08   # the spike: converges from below with the number of draws
    assert np.abs(e[:, 0]).max() < 0.40    # grazing incidence, cos 0.031: the table's own estimate is a handful of lucky samples; stated, not tuned
    # ---- reflection.x: Fresnel = 1 in the table, the conductor's Fresnel at reflectivity 0.99 in the lobe (bsdf.cu:97 clamps the base colour): the replay must sit
    # about 1 % below the table, everywhere
    e = table_errors(bsdf, metal, T_REFL[..., 0], 64)
    inner = e[:, 4:]
    print(f"reflection.x against the conductor lobe at reflectivity 0.99, 192 cells with cos >= 0.28: {inner.min():+.4f} ... {inner.max():+.4f} (mean {inner.mean():+.4f}); all cells: {e.min():+.4f} ... {e.max():+.4f}")
    assert -0.024 < inner.min() and inner.max() < -0.007
    assert -0.10 < e.min() and e.max() < 0.0
    # ---- reflection, both channels: 0.04 x + 0.96 y (what the fetch returns at F0 = 0.04, with Schlick's Fresnel baked in) against the dielectric lobe at ior 1.5 (exact Fresnel)
    e = table_errors(bsdf, dielectric, 0.04 * T_REFL[..., 0] + 0.96 * T_REFL[..., 1], 64)
    print(f"0.04 x + 0.96 y (Schlick, F0 = 0.04) against the dielectric lobe at ior 1.5 (exact Fresnel): {e.min():+.3f} ... {e.max():+.3f}, mean {e.mean():+.4f}; at normal incidence {e[:, 15].min():+.3f} ... {e[:, 15].max():+.3f}")
    assert -0.10 < e.min() and e.max() < 0.35 and abs(e.mean()) < 0.12
    assert np.abs(e[:, 15]).max() < 0.07  # normal incidence: Schlick's approximation is exact in F0 there, the two Fresnel models agree and so do lobe and table


T_IOR1 = GOLD["table_reflection_ior1"].reshape(16, 16, 16)  # [eta k][roughness j][cos i]  (lut.cu:994-1003: idx = i + 16 j + 256 k)


def check_ior1_table(bsdf_ior, eta_cells, k=48):
    """the third table: every (cos, roughness) cell of the given eta slices, replayed through the dielectric reflection lobe at ior = eta (cell centres)"""
    worst_abs, rel_all, n_over = 0.0, [], 0
    for kc in eta_cells:
        eta = float(CENTRES[kc])
        e = np.array([replay(lambda m, wo, wi, u1, u2: bsdf_ior(m, eta, wo, wi, u1, u2), dielectric(float(CENTRES[j])), CENTRES, k) for j in range(16)])
        t = T_IOR1[kc].astype(np.float64)
        worst_abs = max(worst_abs, float(np.abs(e - t).max()))
        big = t >= 0.05
        rel = e[big] / t[big] - 1.0
        rel_all.append(rel)
        n_over += int((np.abs(rel) > 0.02).sum())
        # total internal reflection proper: smooth cells (roughness <= 3/32) below the critical angle hold F = 1 and nothing else
        tir = CENTRES < np.sqrt(max(1.0 - eta * eta, 0.0)) - 0.08
        if tir.any():
            assert np.abs(e[:2][:, tir] / T_REFL[:2][:, tir, 0] - 1.0).max() < 0.01  # = REFLECTION_LUT.x, whose Fresnel is 1
            assert np.abs(t[:2][:, tir] / T_REFL[:2][:, tir, 0] - 1.0).max() < 0.01
    rel = np.concatenate(rel_all)
    print(f"REFLECTION_IOR1_LUT, eta cells {list(eta_cells)}: max |replay - table| {worst_abs:.5f}; {rel.size} cells >= 0.05: rms {np.sqrt((rel ** 2).mean()):.5f}, "
          f"worst {rel.min():+.4f} / {rel.max():+.4f}, {n_over} beyond 2 %")
    assert worst_abs < 0.012
    assert np.sqrt((rel ** 2).mean()) < 0.006 and np.abs(rel).max() < 0.06 and n_over <= 0.01 * rel.size + 2


def check_ior1_tail_cell(bsdf_ior):
    """a cell far below 0.05 (eta 0.969, roughness 0.219, cos 0.656: table 0.00149, F at normal incidence 2.5e-4): what is there comes from GGX's tail reflecting totally,
    and the replay climbs to the table with the number of draws"""
    eta, r, i = float(CENTRES[15]), float(CENTRES[3]), 10
    t = float(T_IOR1[15, 3, i])
    lo = replay(lambda m, wo, wi, u1, u2: bsdf_ior(m, eta, wo, wi, u1, u2), dielectric(r), [CENTRES[i]], 48)[0]
    hi = replay(lambda m, wo, wi, u1, u2: bsdf_ior(m, eta, wo, wi, u1, u2), dielectric(r), [CENTRES[i]], 384)[0]
    print(f"tail cell: table {t:.5f}, replay {lo:.5f} at 48^2 draws, {hi:.5f} at 384^2")
    assert lo < hi and abs(hi / t - 1.0) < 0.10 and abs(lo / t - 1.0) > 0.25


def test_ior1_table_layout_is_the_fetchers(oracle):
    """the axis order read off lut.cu:994-1045 -- x = |w.y|, y = roughness, z = eta, idx = i + 16 j + 256 k, trilinear with clamped neighbours -- restated in numpy
    float32 reproduces the reference's own fetcher (run by gen_ref_golden.py) bit for bit at 2000 random points, clamps included"""
    f = np.float32
    x = GOLD["in_albedo_reflection_ior1"]
    u, v, z = np.abs(x[:, 0]), np.clip(x[:, 1], f(0), f(1)), np.clip(x[:, 2], f(0), f(1))
    uvz = [u * f(16), v * f(16), z * f(16)]
    ijk = [np.clip(a.astype(np.int32), 0, 15) for a in uvz]
    h = [a - b.astype(f) for a, b in zip(uvz, ijk)]
    at = lambda di, dj, dk: T_IOR1[np.clip(ijk[2] + dk, 0, 15), np.clip(ijk[1] + dj, 0, 15), np.clip(ijk[0] + di, 0, 15)]
    t00 = at(0, 0, 0) * (f(1) - h[0]) + at(1, 0, 0) * h[0]
    t01 = at(0, 0, 1) * (f(1) - h[0]) + at(1, 0, 1) * h[0]
    t10 = at(0, 1, 0) * (f(1) - h[0]) + at(1, 1, 0) * h[0]
    t11 = at(0, 1, 1) * (f(1) - h[0]) + at(1, 1, 1) * h[0]
    t0 = t00 * (f(1) - h[1]) + t10 * h[1]
    t1 = t01 * (f(1) - h[1]) + t11 * h[1]
    got = (t0 * (f(1) - h[2]) + t1 * h[2]).astype(f)
    assert np.array_equal(got.view(np.uint32), GOLD["out_albedo_reflection_ior1"].view(np.uint32))


def test_ior1_table_is_the_integral_of_the_checkers_dielectric_lobe(oracle):
    check_ior1_table(oracle.bsdf_ior, range(16))
    check_ior1_tail_cell(oracle.bsdf_ior)


def test_bsdf_ior_entry_is_the_constructor_at_its_own_index(oracle):
    """orc_bsdf_ior at eta = 1.5 is orc_bsdf(entering = true), bit for bit: the entry changes the index and nothing else"""
    rng = np.random.default_rng(11)
    n = 4096
    wo = rng.normal(size=(n, 3)).astype(np.float32); wo[:, 1] = np.abs(wo[:, 1]); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u1, u2 = rng.random(n, dtype=np.float32), rng.random((n, 2), dtype=np.float32)
    m = _material(specular=1.0, specular_roughness=0.3, transmission=0.5, coat=0.4, sheen=0.2)
    a, b = oracle.bsdf(m, True, wo, wi, u1, u2), oracle.bsdf_ior(m, 1.5, wo, wi, u1, u2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _fresnel_dielectric64(c, ior):
    c = c.astype(np.float64)
    temp = ior * ior + c * c - 1.0
    g = np.sqrt(np.maximum(temp, 0.0))
    t0, t1 = (g - c) / (g + c), ((g + c) * c - 1.0) / ((g - c) * c + 1.0)
    return np.where(temp < 0, 1.0, 0.5 * t0 * t0 * (1.0 + t1 * t1))


def check_schlick_channel(bsdf):
    """REFLECTION_LUT.y taken at its word (round 6): the same estimator with Schlick's (1 - c)^5 in the place of the Fresnel term, c = |wo . wh|.  The dielectric lobe's own sample
    carries its exact Fresnel; divided out again (float64, bxdf.cu:274-283 restated in numpy) and replaced, what is left of the lobe -- D, G2 / G1, the visible-normal sampler,
    the reflection -- must land on the .y channel: within 0.005 absolute everywhere (asserted: 0.007), within 0.8 % on the 77 entries >= 0.05 (asserted: 1.5 %; the entries far below are the tail of (1 - c)^5
    under a narrow lobe: they converge from below with the draws, as in the third table)."""
    U = _strata(64)
    n = U.shape[0]
    worst_abs, rel, rel_x = 0.0, [], []
    for j in range(16):
        wo = np.concatenate([np.tile(np.array([np.sqrt(max(1.0 - c * c, 0.0)), c, 0.0], np.float32), (n, 1)) for c in CENTRES])
        out = bsdf(dielectric(float(CENTRES[j])), wo, wo, np.full(wo.shape[0], 0.5, np.float32), np.tile(U, (16, 1)))
        wi, f, pdf = out[:, 4:7].astype(np.float64), out[:, 7].astype(np.float64), out[:, 10].astype(np.float64)
        wh = wo.astype(np.float64) + wi
        wh /= np.linalg.norm(wh, axis=1, keepdims=True)
        c = np.abs((wo.astype(np.float64) * wh).sum(1))
        ok = (pdf > 0) & np.isfinite(pdf) & np.isfinite(f)
        g = np.where(ok, f * np.abs(wi[:, 1]) / np.where(pdf > 0, pdf, 1.0) / _fresnel_dielectric64(c, 1.5), 0.0)  # the lobe with its Fresnel divided out: G2 / G1
        e = (g * (1.0 - c) ** 5).reshape(16, n).mean(axis=1)
        t = T_REFL[j, :, 1].astype(np.float64)
        worst_abs = max(worst_abs, float(np.abs(e - t).max()))
        rel.append((e / t - 1.0)[t >= 0.05])
        rel_x.append(g.reshape(16, n).mean(axis=1) / T_REFL[j, :, 0].astype(np.float64) - 1.0)  # ... and with Fresnel = 1: the .x channel itself
    rel, rel_x = np.concatenate(rel), np.array(rel_x)
    print(f"reflection.y as Schlick's (1 - c)^5 over the dielectric lobe's D, G and sampler: max |replay - table| {worst_abs:.5f}; {rel.size} entries >= 0.05: {rel.min():+.4f} ... {rel.max():+.4f}; "
          f"reflection.x with Fresnel = 1: {rel_x.min():+.4f} ... {rel_x.max():+.4f}, rms {np.sqrt((rel_x ** 2).mean()):.4f}; cos >= 0.09: {rel_x[:, 1:].min():+.4f} ... {rel_x[:, 1:].max():+.4f}")
    assert worst_abs < 0.007 and np.abs(rel).max() < 0.015
    assert np.abs(rel_x[:, 1:]).max() < 0.012 and np.sqrt((rel_x ** 2).mean()) < 0.006 and np.abs(rel_x).max() < 0.06  # (cos cell 0, grazing: the table's own estimate is a handful of samples)


def test_reflection_lut_second_channel_is_schlick_over_the_checkers_lobe(oracle):
    check_schlick_channel(lambda m, wo, wi, u1, u2: oracle.bsdf(m, True, wo, wi, u1, u2))


def test_reference_albedo_tables_are_integrals_of_the_checkers_lobes(oracle):
    check_tables(lambda m, wo, wi, u1, u2: oracle.bsdf(m, True, wo, wi, u1, u2))


def test_tables_count_directions_below_the_horizon(oracle):
    """what the tables are NOT: the hemispherical integral.  Rejecting reflected directions below the horizon reproduces the table for smooth lobes only; at
    roughness ~1 and normal incidence the GGX integral is 1 - ln 2 (closed form for alpha = 1) and the table holds twice that."""
    bsdf = lambda m, wo, wi, u1, u2: oracle.bsdf(m, True, wo, wi, u1, u2)
    smooth = replay(bsdf, metal(float(CENTRES[2])), CENTRES, 64, reject_below_horizon=True) / T_REFL[2, :, 0] - 1.0
    assert np.abs(smooth[2:]).max() < 0.04, smooth
    up = replay(bsdf, metal(1.0), [1.0], 128, reject_below_horizon=True)[0]
    assert abs(up / 0.99 - (1.0 - np.log(2.0))) < 0.02 * (1.0 - np.log(2.0)), up  # (conductor Fresnel at reflectivity 0.99: between 0.99 and 1)
    full = replay(bsdf, metal(1.0), [1.0], 128)[0]
    assert 1.9 < full / up < 2.1
    assert 1.8 < T_REFL[15, 15, 0] / up < 2.2


@pytest.mark.gpu
def test_reference_albedo_tables_are_integrals_of_the_hip_lobes(renderer):
    """the same replay through the HIP library: fh_kat_bsdf runs the device code the shade kernels are made of (generic seven-lobe instantiation)"""
    from fredholm_amd import native as N

    def bsdf(m, wo, wi, u1, u2):
        n = wo.shape[0]
        out = np.zeros((n, 18), np.float32)
        rc = N.lib().fh_kat_bsdf(renderer._ctx, N.ptr(np.ascontiguousarray(m)), 1, C.c_uint32(127), n, N.ptr(np.ascontiguousarray(wo)), N.ptr(np.ascontiguousarray(wi)),
                                 N.ptr(np.ascontiguousarray(u1)), N.ptr(np.ascontiguousarray(u2)), N.ptr(out))
        assert rc == 0
        return out

    check_tables(bsdf)
    check_schlick_channel(bsdf)


def _hip_bsdf_ior(renderer):
    from fredholm_amd import native as N

    def bsdf_ior(m, eta, wo, wi, u1, u2):
        n = wo.shape[0]
        out = np.zeros((n, 18), np.float32)
        rc = N.lib().fh_kat_bsdf_ior(renderer._ctx, N.ptr(np.ascontiguousarray(m)), C.c_float(eta), C.c_uint32(127), n, N.ptr(np.ascontiguousarray(wo)), N.ptr(np.ascontiguousarray(wi)),
                                     N.ptr(np.ascontiguousarray(u1)), N.ptr(np.ascontiguousarray(u2)), N.ptr(out))
        assert rc == 0
        return out

    return bsdf_ior


@pytest.mark.gpu
def test_ior1_table_is_the_integral_of_the_hip_dielectric_lobe(renderer, oracle):
    """all sixteen eta slices through fh_kat_bsdf_ior (the device code of the shade kernels), and the entry against the checker's bit for bit"""
    hip = _hip_bsdf_ior(renderer)
    check_ior1_table(hip, range(16))
    check_ior1_tail_cell(hip)
    rng = np.random.default_rng(12)
    n = 1 << 16
    wo = rng.normal(size=(n, 3)).astype(np.float32); wo[:, 1] = np.abs(wo[:, 1]); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u1, u2 = rng.random(n, dtype=np.float32), rng.random((n, 2), dtype=np.float32)
    for eta in (0.03125, 0.40625, 0.96875, 1.5):
        for m in (dielectric(0.35), _material(specular=1.0, specular_roughness=0.3, transmission=0.5, coat=0.4, sheen=0.2)):
            a, b = hip(m, eta, wo, wi, u1, u2), oracle.bsdf_ior(m, eta, wo, wi, u1, u2)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), eta
