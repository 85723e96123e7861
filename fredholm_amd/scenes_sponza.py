"""A Sponza-class test asset built from scratch (BASELINE.json configs[3]: "Sponza .gltf with textures, 1080p, full Standard Surface").

The Crytek / Intel Sponza meshes are not redistributable and there is no network here, so `sponza_like()` generates an atrium of the
same class procedurally: ~265 k triangles in a glTF node hierarchy (instanced columns, arches, curtains, vases with alpha-cut-out
foliage, two high-resolution bronze heads), 24 textures (base colour, metallic-roughness, normal maps; half of them baseline JPEG with
4:4:4 / 4:2:0 / 4:2:2 subsampling and restart intervals, the others PNG with every filter type; the foliage carries its cut-out in the
base colour's alpha as pt.cu:567-575 expects), clear-coated marble (KHR_materials_clearcoat, scene.cpp:515-530) and metals.
`write_sponza_gltf(path)` writes it the way the reference's loader reads glTF (scene.cpp:445-834: one buffer, ushort indices, float3
POSITION / NORMAL, float2 TEXCOORD_0, external images) and returns the description; both loaders (include/fredholm/scene.h and
fredholm_amd/scene.py) then produce the flat arrays the renderer uploads.
"""
import json
import os

import numpy as np

from . import image_io
from .native import default_materials

F32 = np.float32


# ----------------------------------------------------------------------------------------------- procedural textures (RGBA8, row 0 first)
def _value_noise(h, w, cells, rng):
    """smooth periodic value noise in [0, 1] on an h x w grid"""
    g = rng.random((cells + 1, cells + 1))
    g[-1], g[:, -1] = g[0], g[:, 0]
    y, x = np.mgrid[0:h, 0:w]
    fy, fx = y * (cells / h), x * (cells / w)
    iy, ix = fy.astype(int), fx.astype(int)
    ty, tx = fy - iy, fx - ix
    ty, tx = ty * ty * (3 - 2 * ty), tx * tx * (3 - 2 * tx)
    a, b, c, d = g[iy, ix], g[iy, ix + 1], g[iy + 1, ix], g[iy + 1, ix + 1]
    return (a * (1 - tx) + b * tx) * (1 - ty) + (c * (1 - tx) + d * tx) * ty


def _fbm(h, w, rng, octaves=4, base=4):
    out, amp, tot = np.zeros((h, w)), 1.0, 0.0
    for o in range(octaves):
        out += amp * _value_noise(h, w, base << o, rng)
        tot += amp
        amp *= 0.5
    return out / tot


def _rgba(rgb, alpha=None):
    img = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    img[..., :3] = np.clip(rgb * 255.0 + 0.5, 0, 255).astype(np.uint8)
    img[..., 3] = 255 if alpha is None else np.clip(alpha * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return img


def _normal_map(height, strength):
    """tangent-space normal map (x right, y up in image rows, z out) of a periodic height field"""
    dx = (np.roll(height, -1, axis=1) - np.roll(height, 1, axis=1)) * strength
    dy = (np.roll(height, -1, axis=0) - np.roll(height, 1, axis=0)) * strength
    n = np.stack([-dx, -dy, np.ones_like(height)], axis=2)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    return _rgba(0.5 * n + 0.5)


def _bricks(h, w, rows, cols, rng):
    """(mortar mask in [0, 1], per-brick random tint, height field)"""
    y, x = np.mgrid[0:h, 0:w]
    r = (y * rows) // h
    xo = x + (r % 2) * (w // (2 * cols))
    c = ((xo * cols) // w) % cols
    fy, fx = (y * rows / h) % 1.0, (xo * cols / w) % 1.0
    edge = np.minimum(np.minimum(fx, 1 - fx) * (w / cols), np.minimum(fy, 1 - fy) * (h / rows))
    mortar = np.clip(1.0 - edge / 2.5, 0.0, 1.0)
    tint = rng.random((rows, cols))[r, c]
    return mortar, tint, (1.0 - mortar) * (0.8 + 0.2 * tint)


def make_textures(seed=5):
    """24 RGBA8 textures + the file format each is stored in.  Returns (list of {"rgba8", "srgb", "format", "name"}, name -> index)."""
    rng = np.random.default_rng(seed)
    tex, index = [], {}

    def add(name, img, fmt):
        index[name] = len(tex)
        tex.append({"rgba8": np.ascontiguousarray(img), "srgb": False, "format": fmt, "name": name})  # scene.cpp:564-566: glTF images load NONCOLOR

    n = 256
    # floor: large stone slabs
    mortar, tint, hgt = _bricks(n, n, 8, 4, rng)
    grain = _fbm(n, n, rng)
    col = (0.55 + 0.25 * tint + 0.2 * (grain - 0.5))[..., None] * np.array([1.0, 0.93, 0.82]) * (1 - 0.6 * mortar[..., None])
    add("floor_base", _rgba(col), "jpg")
    add("floor_mr", _rgba(np.stack([np.zeros_like(grain), 0.35 + 0.5 * mortar + 0.15 * grain, np.zeros_like(grain)], 2)), "png")
    add("floor_normal", _normal_map(hgt + 0.15 * grain, 2.0), "png")
    # walls: bricks
    mortar, tint, hgt = _bricks(n, n, 16, 6, rng)
    grain = _fbm(n, n, rng, base=8)
    col = np.stack([0.62 + 0.2 * tint, 0.45 + 0.15 * tint, 0.36 + 0.1 * tint], 2) * (0.85 + 0.3 * (grain[..., None] - 0.5)) * (1 - 0.5 * mortar[..., None]) + 0.35 * mortar[..., None]
    add("wall_base", _rgba(col), "jpg")
    add("wall_normal", _normal_map(hgt + 0.3 * grain, 3.0), "png")
    add("wall_mr", _rgba(np.stack([np.zeros_like(grain), 0.7 + 0.25 * grain, np.zeros_like(grain)], 2)), "jpg")
    # columns: fluted limestone
    y, x = np.mgrid[0:n, 0:n]
    flute = 0.5 + 0.5 * np.cos(x * (2 * np.pi * 16 / n))
    grain = _fbm(n, n, rng, base=6)
    add("column_base", _rgba((0.7 + 0.15 * grain + 0.08 * flute)[..., None] * np.array([1.0, 0.96, 0.88])), "jpg")
    add("column_normal", _normal_map(0.6 * flute + 0.2 * grain, 2.5), "png")
    add("column_mr", _rgba(np.stack([np.zeros_like(grain), 0.45 + 0.3 * grain, np.zeros_like(grain)], 2)), "png")
    # arches: plaster
    grain = _fbm(128, 128, rng, base=8)
    add("arch_base", _rgba((0.78 + 0.15 * grain)[..., None] * np.array([1.0, 0.97, 0.9])), "png")
    # curtains: woven cloth in three colours + one weave normal map
    yy, xx = np.mgrid[0:n, 0:n]
    weave = 0.5 + 0.25 * np.sin(xx * (2 * np.pi * 32 / n)) + 0.25 * np.sin(yy * (2 * np.pi * 32 / n))
    fold = _fbm(n, n, rng, octaves=2, base=2)
    for name, rgb in (("curtain_red", (0.62, 0.06, 0.05)), ("curtain_green", (0.08, 0.42, 0.12)), ("curtain_blue", (0.07, 0.14, 0.55))):
        stripes = 0.8 + 0.2 * ((xx * 12 // n) % 2)
        add(name + "_base", _rgba((0.6 + 0.4 * weave)[..., None] * np.array(rgb) * stripes[..., None] + 0.05 * fold[..., None]), "jpg")
    add("curtain_normal", _normal_map(weave, 1.5), "png")
    # foliage: leaves with a cut-out alpha (PNG keeps the alpha channel)
    m = 128
    yy, xx = np.mgrid[0:m, 0:m]
    alpha = np.zeros((m, m))
    green = np.zeros((m, m, 3))
    for _ in range(26):
        cx, cy, a, b, th = rng.uniform(10, m - 10), rng.uniform(10, m - 10), rng.uniform(8, 18), rng.uniform(3, 7), rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        inside = (u / a) ** 2 + (v / b) ** 2 < 1.0
        alpha[inside] = 1.0
        shade = rng.uniform(0.5, 1.0)
        green[inside] = np.array([0.12, 0.5, 0.1]) * shade + np.array([0.2, 0.25, 0.0]) * (np.abs(v[inside]) < 0.6)[:, None]
    add("leaf_base", _rgba(green, alpha), "png")
    # vases: glazed ceramic
    grain = _fbm(128, 128, rng, base=4)
    band = 0.5 + 0.5 * np.sin(np.mgrid[0:128, 0:128][0] * (2 * np.pi * 6 / 128))
    add("vase_base", _rgba(np.stack([0.15 + 0.5 * band, 0.25 + 0.2 * grain, 0.5 + 0.3 * (1 - band)], 2)), "jpg")
    add("vase_mr", _rgba(np.stack([np.zeros_like(grain), 0.08 + 0.2 * grain, np.zeros_like(grain)], 2)), "png")
    # bronze heads: metal with patina
    pat = _fbm(n, n, rng, octaves=5, base=4)
    patina = np.clip((pat - 0.5) * 4.0, 0.0, 1.0)
    bronze = np.array([0.72, 0.45, 0.2]) * (1 - patina[..., None]) + np.array([0.25, 0.55, 0.45]) * patina[..., None]
    add("bronze_base", _rgba(bronze), "jpg")
    add("bronze_mr", _rgba(np.stack([np.zeros_like(pat), 0.25 + 0.5 * patina, 1.0 - 0.85 * patina], 2)), "png")
    add("bronze_normal", _normal_map(pat, 4.0), "jpg")
    # rug: clear-coated patterned marble inlay in the middle of the floor
    yy, xx = np.mgrid[0:n, 0:n]
    rad = np.hypot(xx - n / 2, yy - n / 2)
    ring = 0.5 + 0.5 * np.cos(rad * (2 * np.pi / 24))
    vein = _fbm(n, n, rng, octaves=5, base=3)
    add("inlay_base", _rgba(np.stack([0.75 * ring + 0.2 * vein, 0.7 * (1 - ring) + 0.25 * vein, 0.55 + 0.3 * vein], 2)), "jpg")
    add("inlay_mr", _rgba(np.stack([np.zeros_like(vein), 0.05 + 0.15 * vein, np.zeros_like(vein)], 2)), "png")
    # balustrade (upper gallery railing): wrought iron + its normal map, and a chain-link cut-out
    grain = _fbm(128, 128, rng, base=8)
    add("iron_base", _rgba((0.2 + 0.15 * grain)[..., None] * np.ones(3)), "png")
    add("iron_mr", _rgba(np.stack([np.zeros_like(grain), 0.35 + 0.3 * grain, 0.9 * np.ones_like(grain)], 2)), "jpg")
    yy, xx = np.mgrid[0:m, 0:m]
    lattice = ((np.abs(((xx + yy) % 32) - 16) < 3) | (np.abs(((xx - yy) % 32) - 16) < 3)).astype(float)
    add("lattice_base", _rgba(np.array([0.25, 0.22, 0.2]) * np.ones((m, m, 3)), lattice), "png")
    # lamps (only used with lamps=True): emission colour pattern
    glow = np.exp(-((xx - m / 2) ** 2 + (yy - m / 2) ** 2) / (2 * 30.0 ** 2))
    add("lamp_emission", _rgba(glow[..., None] * np.array([1.0, 0.85, 0.6])), "png")
    return tex, index


# ----------------------------------------------------------------------------------------------- meshes (local space, indexed)
class Mesh:
    def __init__(self, name):
        self.name = name
        self.parts = []  # (vertices, normals, uvs, triangles, material)

    def add(self, v, n, t, tri, material):
        assert len(v) < 65536, "a glTF primitive of the reference loader addresses vertices with ushort indices (scene.cpp:692-701)"
        self.parts.append((np.asarray(v, F32), np.asarray(n, F32), np.asarray(t, F32), np.asarray(tri, np.uint32), int(material)))

    def n_faces(self):
        return sum(len(p[3]) for p in self.parts)


def _grid(nu, nv, fn, uv_scale=(1.0, 1.0), flip=False):
    """surface p = fn(u, v), u, v in [0, 1]: (nu+1)(nv+1) vertices, smooth normals from central differences, 2 nu nv triangles"""
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    p = fn(u, v)
    e = 1e-4
    du = fn(np.clip(u + e, 0, 1), v) - fn(np.clip(u - e, 0, 1), v)
    dv = fn(u, np.clip(v + e, 0, 1)) - fn(u, np.clip(v - e, 0, 1))
    nrm = np.cross(du, dv)
    if flip:
        nrm = -nrm
    nrm /= np.maximum(np.linalg.norm(nrm, axis=2, keepdims=True), 1e-20)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a, b, c, d = i * (nv + 1) + j, (i + 1) * (nv + 1) + j, (i + 1) * (nv + 1) + j + 1, i * (nv + 1) + j + 1
    tri = np.stack([np.stack([a, b, c], 2), np.stack([a, c, d], 2)], 2).reshape(-1, 3)
    if flip:
        tri = tri[:, ::-1]
    uv = np.stack([u * uv_scale[0], v * uv_scale[1]], 2)
    return p.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), tri


def _lathe(profile, segments, rings, uv_scale=(4.0, 2.0)):
    """surface of revolution about +y of the polyline profile [(radius, y)], resampled to `rings` steps; normals point outwards"""
    prof = np.asarray(profile, float)
    s = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(prof[:, 0]), np.diff(prof[:, 1])))])
    s /= s[-1]

    def fn(u, v):
        r = np.interp(v, s, prof[:, 0])
        y = np.interp(v, s, prof[:, 1])
        ang = 2 * np.pi * u
        return np.stack([r * np.cos(ang), y, -r * np.sin(ang)], 2)

    return _grid(segments, rings, fn, uv_scale)


def build_meshes(mat, detail=1.0):
    """the meshes of the atrium, each in its own local space; mat: material name -> index"""
    d = lambda k: max(2, int(round(k * detail)))
    rng = np.random.default_rng(17)
    meshes = {}

    def mesh(name):
        meshes[name] = Mesh(name)
        return meshes[name]

    # floor: 6.4 x 2.6, slightly uneven; the clear-coated inlay is a second primitive of the same mesh, a hair above it
    bump = lambda u, v: 0.004 * np.sin(37.0 * u) * np.sin(23.0 * v)
    mesh("floor").add(*_grid(d(160), d(64), lambda u, v: np.stack([-3.2 + 6.4 * u, bump(u, v), 1.3 - 2.6 * v], 2), (16.0, 6.0)), mat["floor"])
    meshes["floor"].add(*_grid(d(40), d(40), lambda u, v: np.stack([-0.8 + 1.6 * u, 0.006 + 0 * u, 0.8 - 1.6 * v], 2), (1.0, 1.0)), mat["inlay"])
    # long wall, 6.4 wide x 2.6 high, facing +z, with pilaster relief
    relief = lambda u, v: 0.03 * np.clip(np.cos(u * 2 * np.pi * 9) * 4 - 3, 0, 1) + 0.015 * np.clip(np.cos(v * 2 * np.pi * 2) * 6 - 5, 0, 1)
    mesh("wall_long").add(*_grid(d(200), d(48), lambda u, v: np.stack([-3.2 + 6.4 * u, 2.6 * v, relief(u, v)], 2), (12.0, 5.0)), mat["wall"])
    mesh("wall_short").add(*_grid(d(80), d(48), lambda u, v: np.stack([-1.3 + 2.6 * u, 2.6 * v, relief(u * 0.4, v)], 2), (5.0, 5.0)), mat["wall"])
    # columns: base, shaft with entasis, capital
    lower = [(0.0, 0.0), (0.16, 0.0), (0.16, 0.05), (0.12, 0.08), (0.105, 0.12), (0.1, 0.5), (0.092, 0.95), (0.1, 1.0), (0.13, 1.03), (0.15, 1.08), (0.15, 1.12), (0.0, 1.12)]
    mesh("column_lower").add(*_lathe(lower, d(36), d(32), (4.0, 3.0)), mat["column"])
    upper = [(0.0, 0.0), (0.1, 0.0), (0.1, 0.03), (0.07, 0.06), (0.065, 0.5), (0.06, 0.82), (0.085, 0.86), (0.095, 0.9), (0.0, 0.9)]
    mesh("column_upper").add(*_lathe(upper, d(32), d(24), (3.0, 2.0)), mat["column"])
    # arch between two lower columns: a half ring (span 0.64) extruded 0.24 deep, seen from below and from the front
    def arch(u, v):
        ang = np.pi * u
        r = 0.32 - 0.06 * np.minimum(v * 3.0, 1.0) * (v < 1 / 3) - 0.0 * v
        prof_r = np.where(v < 1 / 3, 0.26 + 0.0 * v, np.where(v < 2 / 3, 0.26 + (v - 1 / 3) * 3 * 0.1, 0.36))
        depth = np.where(v < 1 / 3, -0.12 + v * 3 * 0.24, np.where(v < 2 / 3, 0.12, 0.12 - (v - 2 / 3) * 3 * 0.24))
        return np.stack([-prof_r * np.cos(ang), prof_r * np.sin(ang), depth + 0 * r], 2)
    mesh("arch").add(*_grid(d(32), d(10), arch, (3.0, 1.0), flip=True), mat["arch"])
    # gallery floor slab edge + balustrade with a lattice cut-out (alpha in the base colour)
    mesh("gallery").add(*_grid(d(160), d(6), lambda u, v: np.stack([-3.2 + 6.4 * u, 1.2 - 0.1 * np.cos(np.pi * v) * 0 + 0.08 * (v > 0.5), 0.3 * v], 2), (20.0, 1.0)), mat["arch"])
    meshes["gallery"].add(*_grid(d(120), d(4), lambda u, v: np.stack([-3.2 + 6.4 * u, 1.28 + 0.22 * v, 0.02 + 0 * u], 2), (40.0, 1.0)), mat["lattice"])
    meshes["gallery"].add(*_grid(d(100), d(4), lambda u, v: np.stack([-3.2 + 6.4 * u, 1.5 + 0.015 * np.sin(np.pi * v), 0.0 + 0.03 * v], 2), (30.0, 1.0)), mat["iron"])
    # curtains: hanging cloth 0.9 wide x 1.5 high with folds; one mesh per colour (same geometry, different material)
    def cloth(u, v):
        sag = 0.06 * np.sin(np.pi * u) * (1 - v)
        folds = 0.05 * np.sin(u * 2 * np.pi * 5 + 1.5 * v) * (0.3 + 0.7 * (1 - v))
        return np.stack([-0.45 + 0.9 * u, 1.5 * v - sag, folds], 2)
    for colour in ("red", "green", "blue"):
        mesh("curtain_" + colour).add(*_grid(d(64), d(64), cloth, (3.0, 4.0)), mat["curtain_" + colour])
    # vase (glazed) and its foliage: alpha-cut-out cards around the rim, both windings so that they are lit from either side
    vase = [(0.0, 0.0), (0.09, 0.0), (0.1, 0.02), (0.16, 0.12), (0.17, 0.2), (0.12, 0.3), (0.08, 0.34), (0.1, 0.38), (0.085, 0.38), (0.07, 0.35), (0.0, 0.35)]
    mesh("vase").add(*_lathe(vase, d(32), d(24), (2.0, 1.0)), mat["vase"])
    v_, n_, t_, tri_ = [], [], [], []
    for k in range(d(80)):
        c = np.array([rng.uniform(-0.1, 0.1), rng.uniform(0.36, 0.7), rng.uniform(-0.1, 0.1)])
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        up = np.array([0.0, 1.0, 0.0]) + 0.5 * rng.normal(size=3); up -= ax * np.dot(up, ax); up /= np.linalg.norm(up)
        s = rng.uniform(0.08, 0.16)
        quad = np.array([c - s * ax - s * up, c + s * ax - s * up, c + s * ax + s * up, c - s * ax + s * up])
        nn = np.cross(ax, up)
        for sign, order in ((1.0, (0, 1, 2, 3)), (-1.0, (1, 0, 3, 2))):  # front card and its back side, 0.5 mm apart
            base = len(v_)
            v_ += [quad[i] + sign * 0.00025 * nn for i in order]
            n_ += [sign * nn] * 4
            uvq = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])[list(order)] * 0.5 + rng.integers(0, 2, 2) * 0.5
            t_ += list(uvq)
            tri_ += [[base, base + 1, base + 2], [base, base + 2, base + 3]]
    mesh("foliage").add(np.array(v_), np.array(n_), np.array(t_), np.array(tri_), mat["leaf"])
    # bronze head: a sphere displaced by low-frequency lobes (a stylised lion head), 128 x 96
    def head(u, v):
        th, ph = np.pi * v, 2 * np.pi * u
        r = 0.22 * (1.0 + 0.18 * np.sin(3 * ph) * np.sin(2 * th) ** 2 + 0.1 * np.cos(5 * th) * np.sin(th) + 0.06 * np.sin(9 * ph + 4 * th) * np.sin(th) ** 3
                    + 0.25 * np.exp(-((ph - np.pi) ** 2 + (th - 1.7) ** 2) / 0.08) * np.sin(th))
        return np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th), -r * np.sin(th) * np.sin(ph)], 2)
    mesh("head").add(*_grid(d(128), d(96), head, (2.0, 1.0), flip=True), mat["bronze"])
    # ceiling lamps (optional): small emissive boxes' undersides
    mesh("lamp").add(*_grid(2, 2, lambda u, v: np.stack([-0.12 + 0.24 * u, 0 * u, -0.12 + 0.24 * v], 2), (1.0, 1.0)), mat["lamp"])
    return meshes


def make_materials(tex):
    """material name -> index, and the 180-byte records (glTF metallic-roughness model + clearcoat, as scene.cpp:487-549 fills them)"""
    names = ["floor", "inlay", "wall", "column", "arch", "curtain_red", "curtain_green", "curtain_blue", "leaf", "vase", "bronze", "iron", "lattice", "lamp"]
    mat = {n: i for i, n in enumerate(names)}
    m = default_materials(len(names))
    m["emission"] = 1.0  # the glTF loader sets it (scene.cpp:535-541)

    def setup(name, base=None, mr=None, normal=None, colour=(1.0, 1.0, 1.0), rough=1.0, metal=0.0, coat=0.0, coat_rough=0.0):
        r = m[mat[name]:mat[name] + 1]
        r["base_color"] = colour
        r["specular_roughness"] = rough
        r["metalness"] = metal
        if base: r["base_color_texture_id"] = tex[base]
        if mr: r["metallic_roughness_texture_id"] = tex[mr]
        if normal: r["normalmap_texture_id"] = tex[normal]
        r["coat"], r["coat_roughness"] = coat, coat_rough

    setup("floor", "floor_base", "floor_mr", "floor_normal")
    setup("inlay", "inlay_base", "inlay_mr", None, coat=0.8, coat_rough=0.05)
    setup("wall", "wall_base", "wall_mr", "wall_normal")
    setup("column", "column_base", "column_mr", "column_normal")
    setup("arch", "arch_base", None, None, rough=0.8)
    for c in ("red", "green", "blue"):
        setup("curtain_" + c, f"curtain_{c}_base", None, "curtain_normal", rough=0.9)
    setup("leaf", "leaf_base", None, None, rough=0.6)
    setup("vase", "vase_base", "vase_mr", None)
    setup("bronze", "bronze_base", "bronze_mr", "bronze_normal", metal=1.0)
    setup("iron", "iron_base", "iron_mr", None, metal=1.0)
    setup("lattice", "lattice_base", None, None, rough=0.5, metal=0.0)
    setup("lamp", None, None, None, colour=(0.8, 0.8, 0.8))
    m["emission_color"][mat["lamp"]] = (40.0, 32.0, 20.0)
    m["emission_texture_id"][mat["lamp"]] = tex["lamp_emission"]
    return mat, m


def _quat_y(deg):
    a = np.radians(deg) / 2
    return [0.0, float(np.sin(a)), 0.0, float(np.cos(a))]


def build_nodes(mesh_index, lamps=False):
    """glTF node list (TRS at several levels, instanced meshes, a camera node) and the scene's root nodes"""
    nodes = []

    def node(**kw):
        nodes.append(kw)
        return len(nodes) - 1

    def group(name, children, **kw):
        return node(name=name, children=children, **kw)

    arch_children = [node(name="floor", mesh=mesh_index["floor"])]
    arch_children.append(node(name="wall_north", mesh=mesh_index["wall_long"], translation=[0.0, 0.0, -1.3]))
    arch_children.append(node(name="wall_south", mesh=mesh_index["wall_long"], translation=[0.0, 0.0, 1.3], rotation=_quat_y(180.0)))
    arch_children.append(node(name="wall_west", mesh=mesh_index["wall_short"], translation=[-3.2, 0.0, 0.0], rotation=_quat_y(90.0)))
    arch_children.append(node(name="wall_east", mesh=mesh_index["wall_short"], translation=[3.2, 0.0, 0.0], rotation=_quat_y(-90.0)))
    for side, z, rot in (("north", -0.78, 0.0), ("south", 0.78, 180.0)):
        cols = []
        for k in range(9):
            x = -2.56 + 0.64 * k
            cols.append(node(name=f"column_{side}_{k}", mesh=mesh_index["column_lower"], translation=[x, 0.0, 0.0], rotation=_quat_y(11.0 * k)))
            cols.append(node(name=f"column_up_{side}_{k}", mesh=mesh_index["column_upper"], translation=[x, 1.5, 0.0], scale=[1.0, 1.15, 1.0]))
            if k < 8:
                cols.append(node(name=f"arch_{side}_{k}", mesh=mesh_index["arch"], translation=[x + 0.32, 0.88, 0.0]))
        cols.append(node(name=f"gallery_{side}", mesh=mesh_index["gallery"], translation=[0.0, 0.0, -0.12]))
        arch_children.append(group(f"colonnade_{side}", cols, translation=[0.0, 0.0, z], rotation=_quat_y(rot)))
    architecture = group("architecture", arch_children)
    curtains = []
    for k, (colour, x, z, rot) in enumerate((("red", -1.92, -1.18, 0.0), ("green", -0.64, -1.18, 0.0), ("blue", 0.64, -1.18, 0.0), ("red", 1.92, 1.18, 180.0), ("green", 0.64, 1.18, 180.0),
                                             ("blue", -0.64, 1.18, 180.0))):
        curtains.append(node(name=f"curtain_{k}", mesh=mesh_index["curtain_" + colour], translation=[x, 0.0, z], rotation=_quat_y(rot), scale=[1.0, 1.0 + 0.05 * k, 1.0]))
    drapery = group("drapery", curtains, translation=[0.0, 1.32, 0.0])
    plants = []
    for k in range(8):
        x, z = -2.4 + 0.686 * k, (-0.45 if k % 2 else 0.45)
        leaves = node(name=f"foliage_{k}", mesh=mesh_index["foliage"], rotation=_quat_y(47.0 * k), scale=[1.0 + 0.1 * (k % 3), 1.0, 1.0 + 0.1 * (k % 3)])
        plants.append(node(name=f"vase_{k}", mesh=mesh_index["vase"], translation=[x, 0.0, z], scale=[0.9 + 0.05 * (k % 4)] * 3, children=[leaves]))
    garden = group("garden", plants)
    heads = [node(name="head_near", mesh=mesh_index["head"], translation=[0.55, 0.3, -0.3], rotation=_quat_y(60.0), scale=[1.0, 1.1, 0.9]),
             node(name="head_east", mesh=mesh_index["head"], translation=[2.8, 1.0, 0.0], rotation=_quat_y(-90.0), scale=[1.6, 1.6, 1.6])]
    statues = group("statues", heads)
    children = [architecture, drapery, garden, statues]
    if lamps:
        children.append(group("lamps", [node(name=f"lamp_{k}", mesh=mesh_index["lamp"], translation=[-2.0 + 1.33 * k, 2.45, 0.0], rotation=[1.0, 0.0, 0.0, 0.0]) for k in range(4)]))
    root = group("sponza_like", children, scale=[1.0, 1.0, 1.0])
    cam = node(name="camera", camera=0, matrix=camera_matrix_column_major())
    return nodes, [root, cam]


# the classic view down the long axis from the west end, slightly above eye height.  The reference's thin-lens model puts the lens
# 1 / tan(fov / 2) BEHIND the camera origin (camera.cu:24-53), so the origin sits 1.43 inside the west wall's plane + 0.27
SPONZA_CAMERA = dict(origin=(-1.5, 0.85, 0.12), fov=np.radians(70.0), F=800.0, focus=4.0, forward=(1.0, 0.08, -0.03))  # a small aperture: the atrium is 6.4 units long
SPONZA_SUN = (0.25, 1.0, 0.35)


def camera_matrix_column_major():
    from .renderer import look_at_transform
    t = np.asarray(look_at_transform(SPONZA_CAMERA["origin"], SPONZA_CAMERA["forward"]), F32).reshape(3, 4)
    m = np.eye(4, dtype=F32)
    m[:3, :4] = t
    return [float(m[r, c]) for c in range(4) for r in range(4)]


def sponza_like(detail=1.0, lamps=False):
    """(meshes, nodes, roots, materials, textures): everything write_sponza_gltf needs"""
    tex, tindex = make_textures()
    mat, materials = make_materials(tindex)
    meshes = build_meshes(mat, detail)
    names = [n for n in meshes if lamps or n != "lamp"]
    mesh_index = {n: i for i, n in enumerate(names)}
    if not lamps:
        mesh_index["lamp"] = -1
    nodes, roots = build_nodes(mesh_index, lamps)
    return [meshes[n] for n in names], nodes, roots, materials, tex


def write_sponza_gltf(path, detail=1.0, lamps=False):
    """Write the asset (path.gltf + path.bin + one image file per texture) and return a small description (triangle count after
    instancing, texture files).  Image rows are stored top first and texture v as 1 - v, which the loader undoes (scene.cpp:15, :733)."""
    meshes, nodes, roots, materials, tex = sponza_like(detail, lamps)
    base = os.path.splitext(str(path))[0]
    blob = bytearray()
    views, accessors = [], []

    def add(arr, target, ctype, typ, minmax=False):
        while len(blob) % 4:
            blob.append(0)
        raw = np.ascontiguousarray(arr)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": raw.nbytes, **({"target": target} if target else {})})
        blob.extend(raw.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(raw.shape[0]), "type": typ}
        if minmax:
            acc["min"], acc["max"] = [float(x) for x in raw.min(axis=0)], [float(x) for x in raw.max(axis=0)]
        accessors.append(acc)
        return len(accessors) - 1

    gl_meshes = []
    for m in meshes:
        prims = []
        for v, n, t, tri, material in m.parts:
            uv = t.astype(F32).copy()
            uv[:, 1] = F32(1.0) - uv[:, 1]
            prims.append({"attributes": {"POSITION": add(v, 34962, 5126, "VEC3", True), "NORMAL": add(n, 34962, 5126, "VEC3"), "TEXCOORD_0": add(uv, 34962, 5126, "VEC2")},
                          "indices": add(tri.reshape(-1).astype(np.uint16), 34963, 5123, "SCALAR"), "material": material})
        gl_meshes.append({"name": m.name, "primitives": prims})
    images = []
    for k, tx in enumerate(tex):
        top_first = np.asarray(tx["rgba8"])[::-1]
        if tx["format"] == "jpg":
            fn = f"{os.path.basename(base)}_{tx['name']}.jpg"
            image_io.write_jpeg(os.path.join(os.path.dirname(base), fn), top_first[..., :3], quality=92, subsampling=((1, 1), (2, 2), (2, 1))[k % 3], restart_interval=(0, 8, 3)[k % 3])
        else:
            fn = f"{os.path.basename(base)}_{tx['name']}.png"
            image_io.write_png(os.path.join(os.path.dirname(base), fn), top_first, filter_type=k % 5)
        images.append({"uri": fn, "name": tx["name"]})
    gl_mats = []
    for i, m in enumerate(materials):
        pmr = {"baseColorFactor": [float(x) for x in m["base_color"]] + [1.0], "roughnessFactor": float(m["specular_roughness"]), "metallicFactor": float(m["metalness"])}
        g = {"pbrMetallicRoughness": pmr, "emissiveFactor": [float(x) for x in m["emission_color"]]}
        if m["base_color_texture_id"] >= 0: pmr["baseColorTexture"] = {"index": int(m["base_color_texture_id"])}
        if m["metallic_roughness_texture_id"] >= 0: pmr["metallicRoughnessTexture"] = {"index": int(m["metallic_roughness_texture_id"])}
        if m["normalmap_texture_id"] >= 0: g["normalTexture"] = {"index": int(m["normalmap_texture_id"])}
        if m["emission_texture_id"] >= 0: g["emissiveTexture"] = {"index": int(m["emission_texture_id"])}
        if m["coat"] > 0: g["extensions"] = {"KHR_materials_clearcoat": {"clearcoatFactor": float(m["coat"]), "clearcoatRoughnessFactor": float(m["coat_roughness"])}}
        gl_mats.append(g)
    gl_nodes = []
    for nd in nodes:
        nd = dict(nd)
        if nd.get("mesh", 0) == -1:
            nd.pop("mesh")
        gl_nodes.append(nd)
    doc = {"asset": {"version": "2.0", "generator": "fredholm_amd.scenes_sponza"}, "scene": 0, "scenes": [{"nodes": list(roots)}], "nodes": gl_nodes, "meshes": gl_meshes, "materials": gl_mats,
           "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": os.path.basename(base) + ".bin"}], "images": images,
           "textures": [{"source": k} for k in range(len(images))], "cameras": [{"type": "perspective", "perspective": {"yfov": float(SPONZA_CAMERA["fov"]), "znear": 0.01}}],
           "extensionsUsed": ["KHR_materials_clearcoat"]}
    open(base + ".bin", "wb").write(bytes(blob))
    json.dump(doc, open(str(path), "w"), indent=1)
    # triangles after instancing: every node that references a mesh contributes a copy (scene.cpp:692-760)
    per_mesh = [m.n_faces() for m in meshes]
    n_tris = sum(per_mesh[nd["mesh"]] for nd in gl_nodes if "mesh" in nd)
    return {"triangles": int(n_tris), "nodes": len(gl_nodes), "meshes": len(gl_meshes), "textures": len(images), "jpeg": sum(1 for t in tex if t["format"] == "jpg"),
            "png": sum(1 for t in tex if t["format"] == "png"), "materials": len(gl_mats)}
