"""Synthetic scenes: a from-scratch Cornell box and a seeded triangle soup.

Both return the flat arrays `Renderer.load_scene` uploads -- the layout of the reference's Scene
(fredholm/include/fredholm/scene.h:103-135): unshared vertices, uint3 indices, per-face material
ids, 180-byte Material records.  Triangles are wound so the geometric normal cross(v1-v0, v2-v0)
faces the side that is meant to be lit: the reference BSDF is black (and NaN-weighted) when a
surface is seen from behind (bsdf.cu:56-62).
"""
import numpy as np

from .native import default_materials


def _quad(p0, p1, p2, p3):
    """two triangles (p0,p1,p2), (p0,p2,p3); normal = cross(p1-p0, p2-p0)"""
    return [np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p0, p2, p3)]


def _box(center, half, angle_deg):
    """axis-aligned box rotated about y; outward-facing quads, bottom omitted faces kept (6 faces)."""
    c, s = np.cos(np.radians(angle_deg)), np.sin(np.radians(angle_deg))
    rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    hx, hy, hz = half

    def P(x, y, z):
        return rot @ np.array([x * hx, y * hy, z * hz]) + np.asarray(center)

    quads = [
        (P(-1, 1, 1), P(1, 1, 1), P(1, 1, -1), P(-1, 1, -1)),      # top (+y)
        (P(-1, -1, -1), P(1, -1, -1), P(1, -1, 1), P(-1, -1, 1)),  # bottom (-y)
        (P(-1, -1, 1), P(1, -1, 1), P(1, 1, 1), P(-1, 1, 1)),      # front (+z)
        (P(1, -1, -1), P(-1, -1, -1), P(-1, 1, -1), P(1, 1, -1)),  # back (-z)
        (P(1, -1, 1), P(1, -1, -1), P(1, 1, -1), P(1, 1, 1)),      # right (+x)
        (P(-1, -1, -1), P(-1, -1, 1), P(-1, 1, 1), P(-1, 1, -1)),  # left (-x)
    ]
    out = []
    for q in quads:
        out += _quad(*q)
    return out


def _finish(tris, face_mats, materials):
    v = np.asarray(tris, dtype=np.float32).reshape(-1, 3)
    nf = v.shape[0] // 3
    p0, p1, p2 = v[0::3].astype(np.float64), v[1::3].astype(np.float64), v[2::3].astype(np.float64)
    n = np.cross(p1 - p0, p2 - p0)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    normals = np.repeat(n.astype(np.float32), 3, axis=0)
    uv = np.tile(np.asarray([[0, 0], [1, 0], [0, 1]], dtype=np.float32), (nf, 1))
    return {
        "vertices": v, "normals": normals, "texcoords": uv,
        "indices": np.arange(3 * nf, dtype=np.uint32).reshape(nf, 3),
        "material_ids": np.asarray(face_mats, dtype=np.uint32),
        "materials": materials,
    }


def cornell_box(diffuse_only=False):
    """36 triangles: room x in [-1,1], y in [0,2], z in [-1,1] open towards +z, a ceiling light
    (2 emissive triangles = 2 area lights), a short and a tall block.
    Materials: 0 white, 1 red, 2 green, 3 light.  diffuse_only reproduces BASELINE config 1
    (specular = coat = sheen = 0); otherwise the reference defaults (specular 1, roughness 0.2)."""
    tris, mats = [], []

    def add(q, m):
        tris.extend(q)
        mats.extend([m] * (len(q) // 3))

    add(_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)), 0)      # floor, normal +y
    add(_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)), 0)      # ceiling, normal -y
    add(_quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)), 0)    # back wall, normal +z
    add(_quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)), 1)    # left wall (red), normal +x
    add(_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)), 2)        # right wall (green), normal -x
    add(_quad((-0.35, 1.995, -0.3), (0.35, 1.995, -0.3), (0.35, 1.995, 0.3), (-0.35, 1.995, 0.3)), 3)  # light, normal -y
    add(_box((0.38, 0.3, 0.35), (0.3, 0.3, 0.3), -17.0), 0)            # short block
    add(_box((-0.35, 0.6, -0.3), (0.3, 0.6, 0.3), 20.0), 0)            # tall block
    m = default_materials(4)
    m["base_color"][0] = (0.73, 0.73, 0.73)
    m["base_color"][1] = (0.65, 0.05, 0.05)
    m["base_color"][2] = (0.12, 0.45, 0.15)
    m["base_color"][3] = (0.78, 0.78, 0.78)
    m["emission"][3] = 1.0
    m["emission_color"][3] = (17.0, 12.0, 4.0)
    if diffuse_only:
        m["specular"] = 0.0
        m["coat"] = 0.0
        m["sheen"] = 0.0
    return _finish(tris, mats, m)


CORNELL_CAMERA = dict(origin=(0.0, 1.0, 1.0), fov=0.5 * np.pi, F=100.0, focus=10000.0)  # GUI defaults, controller.h:89-92


def pcg32_floats(n, state=0x853C49E6748FEA9B, inc=0xDA3E39CB94B95BDB):
    """n floats in [0,1) from one PCG32 (XSH RR) stream, vectorised: the LCG states are produced
    in closed form with wrapping uint64 cumulative products/sums."""
    a = np.uint64(6364136223846793005)
    c = np.uint64(inc | 1)
    with np.errstate(over="ignore"):
        apow = np.empty(n, dtype=np.uint64)
        apow[0] = 1
        if n > 1:
            apow[1:] = a
            np.cumprod(apow, out=apow)
        geo = np.empty(n, dtype=np.uint64)  # sum_{j<k} a^j
        geo[0] = 0
        if n > 1:
            np.cumsum(apow[:-1], out=geo[1:])
        old = apow * np.uint64(state) + geo * c
    xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
    rot = (old >> np.uint64(59)).astype(np.uint32)
    out = (xorshifted >> rot) | (xorshifted << ((-rot.astype(np.int64)) & 31).astype(np.uint32))
    return ((out >> np.uint32(8)).astype(np.float32)) * np.float32(1.0 / 16777216.0)


SOUP_PALETTE = np.asarray([
    (0.80, 0.80, 0.80), (0.80, 0.25, 0.20), (0.20, 0.60, 0.25), (0.20, 0.35, 0.80),
    (0.85, 0.75, 0.25), (0.70, 0.30, 0.70), (0.25, 0.70, 0.70), (0.95, 0.60, 0.30)], dtype=np.float32)


def triangle_soup(n_tris=1_000_000, edge_scale=0.02):
    """centres U([-1,1]^3), three edge vectors U([-1,1]^3)*edge_scale per
    triangle, face normals, 8 materials cycling (roughness 0.2..0.8, metalness 0/1)."""
    r = pcg32_floats(12 * n_tris).reshape(n_tris, 12) * np.float32(2.0) - np.float32(1.0)
    centre = r[:, 0:3]
    v = np.empty((n_tris, 3, 3), dtype=np.float32)
    for k in range(3):
        v[:, k, :] = centre + r[:, 3 + 3 * k:6 + 3 * k] * np.float32(edge_scale)
    m = default_materials(8)
    for i in range(8):
        m["base_color"][i] = SOUP_PALETTE[i]
        m["specular_roughness"][i] = 0.2 + 0.6 * i / 7.0
        m["metalness"][i] = float(i % 2)
    mats = (np.arange(n_tris) % 8).astype(np.uint32)
    return _finish(v.reshape(-1, 3), mats, m)


SOUP_CAMERA = dict(origin=(0.0, 0.0, 3.0), fov=np.radians(60.0), F=100.0, focus=10000.0)
SOUP_SUN = (-0.1, 1.0, 0.1)  # rtcamp8.cpp:142-146
