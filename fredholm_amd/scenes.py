"""Synthetic scenes: a from-scratch Cornell box and a seeded triangle soup.

Both return the flat arrays `Renderer.load_scene` uploads -- the layout of the reference's Scene
(fredholm/include/fredholm/scene.h:103-135): unshared vertices, uint3 indices, per-face material
ids, 180-byte Material records.  Triangles are wound so the geometric normal cross(v1-v0, v2-v0)
faces the side that is meant to be lit: the reference BSDF is black (and NaN-weighted) when a
surface is seen from behind (bsdf.cu:56-62).
"""
import numpy as np

from . import image_io

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


def soup_with_emitters(n_tris=1_000_000, edge_scale=0.02):
    """BASELINE configs[4] stand-in (the rtcamp8 asset is not in the tree -- "C3 geometry + Cornell emitters"):
    the triangle soup plus the Cornell ceiling panel (2 emissive triangles = 2 area lights, emission (17, 12, 4)) above it, facing down."""
    sc = triangle_soup(n_tris, edge_scale)
    panel = np.asarray(_quad((-0.6, 1.3, -0.6), (0.6, 1.3, -0.6), (0.6, 1.3, 0.6), (-0.6, 1.3, 0.6)), dtype=np.float32)  # normal -y
    nm = sc["materials"].shape[0]
    mats = default_materials(nm + 1)
    mats[:nm] = sc["materials"]
    mats["base_color"][nm] = (0.78, 0.78, 0.78)
    mats["emission"][nm] = 1.0
    mats["emission_color"][nm] = (17.0, 12.0, 4.0)
    verts = np.concatenate([sc["vertices"], panel])
    ids = np.concatenate([sc["material_ids"], np.asarray([nm, nm], np.uint32)])
    return _finish(verts, ids, mats)


SOUP_CAMERA = dict(origin=(0.0, 0.0, 3.0), fov=np.radians(60.0), F=100.0, focus=10000.0)
SOUP_SUN = (-0.1, 1.0, 0.1)  # rtcamp8.cpp:142-146


# ---------------------------------------------------------------------------------------------
# Wavefront .obj/.mtl in and out (the wire format in front of the hot path).
# The reader applies the reference's tinyobjloader mapping (fredholm/src/scene.cpp:119-443): Kd -> base_color,
# Ks -> specular_color, Pr / Pm / Pc, coat_roughness <- clearcoat_thickness (:240-242), transmission = 1 - d (:245),
# Tf, Ke, custom keys diffuse, diffuse_roughness, sheen*, subsurface*, thin_walled; face normals and barycentric
# texcoords when absent (:361-377).  include/fredholm/scene.h is the C++ twin of this reader.
# ---------------------------------------------------------------------------------------------
def write_obj(scene, path):
    """Write a flat scene as .obj + .mtl next to it (normals are regenerated on load; texture coordinates are written when the
    scene has textures, which go next to the .mtl as PNG files, stored bottom row last like any image file)."""
    import os
    base = os.path.splitext(path)[0]
    mats = scene["materials"]
    textures = scene.get("textures") or []
    for k, tex in enumerate(textures):
        image_io.write_png(f"{base}_tex{k}.png", np.asarray(tex["rgba8"])[::-1], filter_type=k % 5)
    inv = {v[0]: k for k, v in _MTL_TEXTURES.items() if k not in ("map_Bump", "bump")}
    with open(base + ".mtl", "w") as f:
        for i, m in enumerate(mats):
            f.write(f"newmtl m{i}\n")
            for field, stmt in inv.items():
                if m[field] >= 0:
                    f.write(f"{stmt} {os.path.basename(base)}_tex{int(m[field])}.png\n")
            f.write("Kd %.9g %.9g %.9g\n" % tuple(m["base_color"]))
            f.write("Ks %.9g %.9g %.9g\n" % tuple(m["specular_color"]))
            f.write("Pr %.9g\nPm %.9g\n" % (m["specular_roughness"], m["metalness"]))
            if m["coat"] > 0:
                f.write("Pc %.9g\n" % m["coat"])
            f.write("d %.9g\n" % (1.0 - m["transmission"]))
            if (m["emission_color"] > 0).any():
                f.write("Ke %.9g %.9g %.9g\n" % tuple(m["emission_color"]))
            f.write("diffuse %.9g\ndiffuse_roughness %.9g\n" % (m["diffuse"], m["diffuse_roughness"]))
            f.write("sheen %.9g\nsheen_color %.9g %.9g %.9g\nsheen_roughness %.9g\n" % (m["sheen"], *m["sheen_color"], m["sheen_roughness"]))
            f.write("subsurface %.9g\nsubsurface_color %.9g %.9g %.9g\nthin_walled %.9g\n" % (m["subsurface"], *m["subsurface_color"], m["thin_walled"]))
            f.write("specular_weight %.9g\n\n" % m["specular"])
    v = scene["vertices"]
    with open(path, "w") as f:
        f.write(f"mtllib {os.path.basename(base)}.mtl\n")
        for p in v:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        if textures:
            for t in scene["texcoords"]:
                f.write("vt %.9g %.9g\n" % tuple(t))
        cur = None
        for face, mid in zip(scene["indices"], scene["material_ids"]):
            if mid != cur:
                f.write(f"usemtl m{mid}\n")
                cur = mid
            if textures:
                f.write("f %d/%d %d/%d %d/%d\n" % tuple(int(i) + 1 for i in face for _ in (0, 1)))
            else:
                f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in face))


def load_obj(path):
    """Minimal .obj/.mtl reader producing the flat arrays Renderer.load_scene takes."""
    import os
    pos, nrm, tex = [], [], []
    mats, mat_index = [], {}
    verts, norms, uvs, faces, mids = [], [], [], [], []
    textures, tex_index = [], {}
    cur = -1

    def load_mtl(p):
        m = None
        pc = 0.0
        for line in open(p):
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                m = default_materials(1)[0]
                m["base_color"] = 0.0
                m["specular_color"] = 0.0
                mat_index[t[1]] = len(mats)
                mats.append(m)
                pc = 0.0
                continue
            if m is None:
                continue
            f = [float(x) for x in t[1:]] if all(_isnum(x) for x in t[1:]) else None
            if t[0] == "Kd": m["base_color"] = f
            elif t[0] == "Ks": m["specular_color"] = f
            elif t[0] == "Pr" and f[0] > 0: m["specular_roughness"] = f[0]
            elif t[0] == "Pm": m["metalness"] = f[0]
            elif t[0] == "Pc":
                pc = f[0]
                if pc > 0: m["coat"] = pc
            elif t[0] == "Pcr" and f[0] > 0: m["coat_roughness"] = pc
            elif t[0] == "d": m["transmission"] = max(np.float32(1.0) - np.float32(f[0]), 0.0)
            elif t[0] == "Tf" and max(f) > 0: m["transmission_color"] = f
            elif t[0] == "Ke" and max(f) > 0:
                m["emission"] = 1.0
                m["emission_color"] = f
            elif t[0] in ("diffuse", "diffuse_roughness", "sheen", "sheen_roughness", "subsurface", "thin_walled"): m[t[0]] = f[0]
            elif t[0] in ("sheen_color", "subsurface_color"): m[t[0]] = f
            elif t[0] in _MTL_TEXTURES:
                # scene.cpp:144-153,196-310: one texture per distinct file name (its first use fixes COLOR / NONCOLOR); the last
                # token is the file name (options such as "-bm 1" precede it)
                field, srgb = _MTL_TEXTURES[t[0]]
                name = t[-1]
                if name not in tex_index:
                    tex_index[name] = len(textures)
                    textures.append({"rgba8": image_io.load_texture(os.path.join(os.path.dirname(p), name), flip_vertically=True), "srgb": srgb})
                m[field] = tex_index[name]
            elif t[0].startswith("map_"):
                raise ValueError(f"{t[0]} is not a texture slot of the reference's .mtl mapping ({p})")

    for line in open(path):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v": pos.append([float(x) for x in t[1:4]])
        elif t[0] == "vn": nrm.append([float(x) for x in t[1:4]])
        elif t[0] == "vt": tex.append([float(x) for x in t[1:3]])
        elif t[0] == "mtllib": load_mtl(os.path.join(os.path.dirname(path), t[1]))
        elif t[0] == "usemtl": cur = mat_index.get(t[1], -1)
        elif t[0] == "f":
            cs = []
            for tok in t[1:]:
                parts = (tok.split("/") + ["", ""])[:3]
                cs.append(tuple(int(x) if x else 0 for x in parts))
            for k in range(1, len(cs) - 1):
                tri = (cs[0], cs[k], cs[k + 1])
                p = [np.asarray(pos[c[0] - 1 if c[0] > 0 else len(pos) + c[0]], dtype=np.float32) for c in tri]
                has_n, has_t = all(c[2] for c in tri), all(c[1] for c in tri)
                if not has_n:  # scene.cpp:361-371, in single precision operation by operation (the C++ reader does the same)
                    e1, e2 = _normalize32(p[1] - p[0]), _normalize32(p[2] - p[0])
                    fn = _normalize32(np.asarray([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]], dtype=np.float32))
                base = len(verts)
                for j, c in enumerate(tri):
                    verts.append(p[j])
                    norms.append(nrm[c[2] - 1 if c[2] > 0 else len(nrm) + c[2]] if has_n else fn)
                    uvs.append(tex[c[1] - 1 if c[1] > 0 else len(tex) + c[1]] if has_t else [(0, 0), (1, 0), (0, 1)][j])
                faces.append((base, base + 1, base + 2))
                mids.append(cur)
    if any(m < 0 for m in mids):
        mats.append(default_materials(1)[0])
        mids = [len(mats) - 1 if m < 0 else m for m in mids]
    out = {"vertices": np.asarray(verts, dtype=np.float32), "normals": np.asarray(norms, dtype=np.float32), "texcoords": np.asarray(uvs, dtype=np.float32),
           "indices": np.asarray(faces, dtype=np.uint32), "material_ids": np.asarray(mids, dtype=np.uint32), "materials": np.asarray(mats, dtype=default_materials(1).dtype)}
    if textures:
        out["textures"] = textures
    return out


# .mtl statement -> (Material field, sRGB) as tinyobjloader names them and scene.cpp:196-310 consumes them
_MTL_TEXTURES = {"map_Kd": ("base_color_texture_id", True), "map_Ks": ("specular_color_texture_id", True), "map_Pr": ("specular_roughness_texture_id", False),
                 "map_Pm": ("metalness_texture_id", False), "map_bump": ("heightmap_texture_id", False), "map_Bump": ("heightmap_texture_id", False),
                 "bump": ("heightmap_texture_id", False), "norm": ("normalmap_texture_id", False), "map_d": ("alpha_texture_id", False)}


def _normalize32(v):
    x, y, z = np.float32(v[0]), np.float32(v[1]), np.float32(v[2])
    l = np.sqrt(np.float32(np.float32(x * x + y * y) + z * z))
    return np.asarray([x / l, y / l, z / l], dtype=np.float32)


def _isnum(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def checker_texture(w, h, cells, c0, c1, alpha0=255, alpha1=255, seed=0):
    """RGBA8 checkerboard with a little per-texel noise (so that filtering matters); row 0 first."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    on = (((x * cells) // w + (y * cells) // h) % 2).astype(bool)
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[..., :3] = np.where(on[..., None], np.asarray(c1, dtype=np.uint8), np.asarray(c0, dtype=np.uint8))
    img[..., :3] = np.clip(img[..., :3].astype(np.int32) + rng.integers(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)
    img[..., 3] = np.where(on, alpha1, alpha0)
    return img


def textured_cornell_box():
    """Cornell box exercising every texture slot of the reference (pt.cu:181-280, :545-678, :709-742, :131-139):
    sRGB base-colour checker on the floor, alpha cut-out card, normal-mapped back wall, height-mapped left wall,
    roughness/metalness maps on the short block, coat maps on the tall block, textured emitter."""
    sc = cornell_box()
    m = default_materials(9)
    m[:4] = sc["materials"]
    rng = np.random.default_rng(7)
    tex = []

    def add(img, srgb):
        tex.append({"rgba8": img, "srgb": srgb})
        return len(tex) - 1

    # 4: floor with base-colour checker (opaque alpha)
    m["base_color_texture_id"][4] = add(checker_texture(64, 64, 8, (200, 60, 40), (230, 230, 210)), True)
    # 5: back wall with normal map
    nm = np.zeros((32, 32, 4), np.uint8)
    nm[..., 0] = (128 + 60 * np.sin(np.arange(32) / 32 * 6 * np.pi))[None, :].astype(np.uint8)
    nm[..., 1] = 128
    nm[..., 2] = 230
    nm[..., 3] = 255
    m["normalmap_texture_id"][5] = add(nm, False)
    m["base_color"][5] = (0.7, 0.7, 0.75)
    # 6: left wall with height map
    hm = np.zeros((48, 24, 4), np.uint8)
    hm[..., 0] = rng.integers(0, 256, (48, 24))
    hm[..., 3] = 255
    m["heightmap_texture_id"][6] = add(hm, False)
    m["base_color"][6] = (0.65, 0.05, 0.05)
    # 7: short block: roughness + metalness + specular colour maps, tall block (8): coat + coat roughness + glTF metallic-roughness
    m["specular_roughness_texture_id"][7] = add(checker_texture(16, 16, 4, (40, 0, 0), (200, 0, 0), seed=1), False)
    m["metalness_texture_id"][7] = add(checker_texture(16, 16, 2, (0, 0, 0), (255, 0, 0), seed=2), False)
    m["specular_color_texture_id"][7] = add(checker_texture(8, 8, 2, (255, 255, 255), (255, 200, 120), seed=3), True)
    m["base_color"][7] = (0.8, 0.7, 0.3)
    m["coat_texture_id"][8] = add(checker_texture(16, 16, 4, (30, 0, 0), (255, 0, 0), seed=4), False)
    m["coat_roughness_texture_id"][8] = add(checker_texture(16, 16, 4, (0, 20, 0), (0, 120, 0), seed=5), False)
    m["metallic_roughness_texture_id"][8] = add(checker_texture(16, 16, 2, (0, 60, 0), (0, 160, 255), seed=6), False)
    # light with emission texture
    em = checker_texture(8, 8, 2, (255, 255, 255), (255, 120, 40), seed=8)
    m["emission_texture_id"][3] = add(em, True)
    ids = sc["material_ids"].copy()
    ids[0:2] = 4     # floor
    ids[4:6] = 5     # back wall
    ids[6:8] = 6     # left wall
    ids[12:24] = 7   # short block
    ids[24:36] = 8   # tall block
    # alpha cut-out card in front of the back wall: base-colour alpha on one triangle pair, alpha texture on another
    card = _quad((-0.6, 0.9, -0.5), (0.2, 0.9, -0.5), (0.2, 1.7, -0.5), (-0.6, 1.7, -0.5)) + _quad((0.3, 0.9, -0.4), (0.9, 0.9, -0.4), (0.9, 1.5, -0.4), (0.3, 1.5, -0.4))
    mc = default_materials(2)
    mc["base_color_texture_id"][0] = add(checker_texture(32, 32, 4, (40, 200, 60), (0, 0, 0), alpha0=255, alpha1=0, seed=9), True)
    mc["alpha_texture_id"][1] = add(checker_texture(32, 32, 6, (255, 0, 0), (0, 0, 0), seed=10), False)
    mc["base_color"][1] = (0.2, 0.3, 0.8)
    verts = np.concatenate([sc["vertices"], np.asarray(card, dtype=np.float32)])
    nf = verts.shape[0] // 3
    out = _finish(verts, np.concatenate([ids, [9, 9, 10, 10]]).astype(np.uint32), np.concatenate([m, mc]))
    # per-face uv: stretch the unit triangle uvs a bit so that wrap addressing is exercised
    out["texcoords"] = (out["texcoords"] * np.float32(1.7) - np.float32(0.2)).astype(np.float32)
    out["textures"] = tex
    assert out["indices"].shape[0] == nf
    return out


def gradient_ibl(w=64, h=32):
    """small lat-long float environment: warm sun blob + blue-ish gradient"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), dtype=np.float32)
    img[..., 0] = 0.3 + 0.4 * (1 - y / h)
    img[..., 1] = 0.4 + 0.4 * (1 - y / h)
    img[..., 2] = 0.6 + 0.6 * (1 - y / h)
    blob = np.exp(-(((x - 0.7 * w) / 4) ** 2 + ((y - 0.25 * h) / 3) ** 2))
    img[..., :3] += 30.0 * blob[..., None] * np.asarray([1.0, 0.85, 0.6], dtype=np.float32)
    img[..., 3] = 1.0
    return img


def write_gltf(scene, path, submeshes, nodes, roots, animations=(), cameras=0, embed=False, image_format="png"):
    """Write a flat scene as glTF 2.0 the way the reference's loader wants it (scene.cpp:692-741: 16-bit indices, float3
    POSITION / NORMAL, float2 TEXCOORD_0, one buffer).  For tests and tools.
      submeshes: list of face-index arrays, one glTF mesh each; a mesh gets one primitive per material it uses
      nodes:     list of dicts with optional keys mesh, translation, rotation (x, y, z, w), scale, matrix (16, column-major),
                 children, camera
      roots:     node indices of scene 0
      animations: list of channel lists; a channel = (node, path, key times, key values)
    Materials carry base colour, roughness, metalness, emissive factor and the base-colour / normal / emission /
    metallic-roughness textures (written next to the file as PNG, top row first, so that the loader's v -> 1 - v plus its
    vertical image flip land on the same texels)."""
    import base64
    import json
    import os
    base = os.path.splitext(path)[0]
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
            acc["min"], acc["max"] = [float(v) for v in np.atleast_1d(raw.min(axis=0))], [float(v) for v in np.atleast_1d(raw.max(axis=0))]
        accessors.append(acc)
        return len(accessors) - 1

    v, n, t, idx, mid = scene["vertices"], scene["normals"], scene["texcoords"], scene["indices"], scene["material_ids"]
    meshes = []
    for faces in submeshes:
        prims = []
        faces = np.asarray(faces, dtype=np.int64)
        for m in sorted(set(int(x) for x in mid[faces])):
            f = faces[mid[faces] == m]
            vid = idx[f].reshape(-1)
            assert len(vid) < 65536
            uv = t[vid].astype(np.float32).copy()
            uv[:, 1] = np.float32(1.0) - uv[:, 1]  # undone by the loader (scene.cpp:733)
            prims.append({"attributes": {"POSITION": add(v[vid].astype(np.float32), 34962, 5126, "VEC3", True), "NORMAL": add(n[vid].astype(np.float32), 34962, 5126, "VEC3"),
                                         "TEXCOORD_0": add(uv, 34962, 5126, "VEC2")},
                          "indices": add(np.arange(len(vid), dtype=np.uint16), 34963, 5123, "SCALAR"), "material": m})
        meshes.append({"primitives": prims})
    anims = []
    for channels in animations:
        samplers, chans = [], []
        for node, pth, times, values in channels:
            vals = np.asarray(values, dtype=np.float32)
            samplers.append({"input": add(np.asarray(times, dtype=np.float32), None, 5126, "SCALAR", True), "output": add(vals, None, 5126, "VEC4" if pth == "rotation" else "VEC3"),
                             "interpolation": "LINEAR"})
            chans.append({"sampler": len(samplers) - 1, "target": {"node": node, "path": pth}})
        anims.append({"samplers": samplers, "channels": chans})
    textures = scene.get("textures") or []
    images = []
    for k, tex in enumerate(textures):
        top_first = np.asarray(tex["rgba8"])[::-1]
        if image_format == "jpg" and k % 2 == 0:  # every other texture as baseline JPEG (lossy: the loader sees what the file decodes to)
            image_io.write_jpeg(f"{base}_img{k}.jpg", top_first[..., :3], quality=90, subsampling=((1, 1), (2, 2), (2, 1))[k // 2 % 3], restart_interval=k % 4)
            images.append({"uri": f"{os.path.basename(base)}_img{k}.jpg"})
        else:
            image_io.write_png(f"{base}_img{k}.png", top_first, filter_type=(k + 1) % 5)
            images.append({"uri": f"{os.path.basename(base)}_img{k}.png"})
    mats = []
    for m in scene["materials"]:
        pmr = {"baseColorFactor": [float(x) for x in m["base_color"]] + [1.0], "roughnessFactor": float(m["specular_roughness"]), "metallicFactor": float(m["metalness"])}
        g = {"pbrMetallicRoughness": pmr, "emissiveFactor": [float(x) for x in m["emission_color"]]}
        if m["base_color_texture_id"] >= 0: pmr["baseColorTexture"] = {"index": int(m["base_color_texture_id"])}
        if m["metallic_roughness_texture_id"] >= 0: pmr["metallicRoughnessTexture"] = {"index": int(m["metallic_roughness_texture_id"])}
        if m["normalmap_texture_id"] >= 0: g["normalTexture"] = {"index": int(m["normalmap_texture_id"])}
        if m["emission_texture_id"] >= 0: g["emissiveTexture"] = {"index": int(m["emission_texture_id"])}
        if m["coat"] > 0: g["extensions"] = {"KHR_materials_clearcoat": {"clearcoatFactor": float(m["coat"]), "clearcoatRoughnessFactor": float(m["coat_roughness"])}}
        mats.append(g)
    doc = {"asset": {"version": "2.0", "generator": "fredholm_amd.scenes.write_gltf"}, "scene": 0, "scenes": [{"nodes": list(roots)}], "nodes": [dict(nd) for nd in nodes], "meshes": meshes,
           "materials": mats, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": len(blob), "uri": ("data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()) if embed else os.path.basename(base) + ".bin"}]}
    if anims: doc["animations"] = anims
    if images:
        doc["images"] = images
        doc["textures"] = [{"source": k} for k in range(len(images))]
    if cameras:
        doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": 1.0, "znear": 0.01}} for _ in range(cameras)]
    if not embed:
        open(base + ".bin", "wb").write(bytes(blob))
    json.dump(doc, open(path, "w"), indent=1)


def animated_cornell_gltf(path, embed=False, textured=True, image_format="png"):
    """Test asset: the Cornell box as a glTF scene graph -- room (root), a key-framed root node carrying the short block with the
    tall block as its child (so the child inherits the animation), a static scaled/rotated node, and a camera node."""
    sc = textured_cornell_box() if textured else cornell_box()
    if textured:
        # keep only what glTF can express (scene.cpp:487-549): base colour, metallic-roughness, normal, emission textures
        for f in ("specular_color_texture_id", "specular_roughness_texture_id", "metalness_texture_id", "coat_texture_id", "coat_roughness_texture_id", "heightmap_texture_id",
                  "alpha_texture_id"):
            sc["materials"][f] = -1
    nf = sc["indices"].shape[0]
    room, short, tall, rest = np.arange(0, 12), np.arange(12, 24), np.arange(24, 36), np.arange(36, nf)
    submeshes = [room, short, tall] + ([rest] if len(rest) else [])
    s = float(np.sin(0.3)), float(np.cos(0.3))
    nodes = [{"mesh": 0, "name": "room"},
             {"mesh": 1, "name": "short", "translation": [0.1, 0.0, 0.05], "children": [2]},
             {"mesh": 2, "name": "tall", "translation": [-0.05, 0.0, 0.1], "rotation": [0.0, s[0], 0.0, s[1]], "scale": [0.9, 1.05, 0.9]},
             {"camera": 0, "name": "cam", "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.05, 1.0, 1.2, 1]}]
    roots = [0, 1, 3]
    if len(rest):
        nodes.append({"mesh": 3, "name": "cards", "scale": [1.0, 1.0, 1.0]})
        roots.append(4)
    h = np.sqrt(0.5)
    animations = [[(1, "translation", [0.0, 0.5, 1.5, 2.0], [[0.1, 0.0, 0.05], [0.2, 0.05, 0.0], [0.0, 0.1, 0.1], [0.1, 0.0, 0.05]]),
                   (1, "rotation", [0.0, 1.0, 2.0], [[0, 0, 0, 1], [0, float(h), 0, float(h)], [0, 1, 0, 0]]),
                   (1, "scale", [0.0, 2.0], [[1, 1, 1], [0.8, 1.2, 0.8]])]]
    write_gltf(sc, path, submeshes, nodes, roots, animations, cameras=1, embed=embed, image_format=image_format)
    return sc


def city(n_blocks=40000, seed=11):
    """A non-uniform scene for builder comparisons: a ground plane of two huge triangles, `n_blocks` boxes of widely varying size on a
    jittered grid (towers next to kerb stones), and a few hundred long thin triangles spanning the whole extent (cables).  Vectorised:
    12 triangles per box."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n_blocks)))
    gx, gz = np.meshgrid(np.arange(side), np.arange(side))
    gx, gz = gx.reshape(-1)[:n_blocks].astype(np.float64), gz.reshape(-1)[:n_blocks].astype(np.float64)
    cell = 2.0 / side
    cx = -1.0 + (gx + 0.5 + rng.uniform(-0.3, 0.3, n_blocks)) * cell
    cz = -1.0 + (gz + 0.5 + rng.uniform(-0.3, 0.3, n_blocks)) * cell
    hx = cell * rng.uniform(0.05, 0.45, n_blocks)
    hz = cell * rng.uniform(0.05, 0.45, n_blocks)
    hy = cell * np.exp(rng.uniform(np.log(0.05), np.log(12.0), n_blocks))
    ang = rng.uniform(0, np.pi, n_blocks)
    c, s = np.cos(ang), np.sin(ang)
    corners = np.array([[-1, -1, -1], [1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]], dtype=np.float64)
    lx, ly, lz = corners[:, 0][None] * hx[:, None], (corners[:, 1][None] + 1.0) * hy[:, None], corners[:, 2][None] * hz[:, None]
    P = np.stack([cx[:, None] + c[:, None] * lx + s[:, None] * lz, ly, cz[:, None] - s[:, None] * lx + c[:, None] * lz], axis=-1)  # [n, 8, 3]
    quads = [(7, 6, 5, 4), (0, 1, 2, 3), (3, 2, 6, 7), (1, 0, 4, 5), (2, 1, 5, 6), (0, 3, 7, 4)]
    tri_idx = np.array([[q[0], q[1], q[2], q[0], q[2], q[3]] for q in quads]).reshape(-1)
    boxes = P[:, tri_idx].reshape(-1, 3)
    ground = np.array(_quad((-1.2, 0, 1.2), (1.2, 0, 1.2), (1.2, 0, -1.2), (-1.2, 0, -1.2)))
    n_cables = 300
    a = np.stack([rng.uniform(-1, 1, n_cables), rng.uniform(0.2, 1.5, n_cables), rng.uniform(-1, 1, n_cables)], axis=1)
    b = np.stack([rng.uniform(-1, 1, n_cables), rng.uniform(0.2, 1.5, n_cables), rng.uniform(-1, 1, n_cables)], axis=1)
    cables = np.stack([a, b, b + np.array([0.0, 0.004, 0.0])], axis=1).reshape(-1, 3)
    tris = np.concatenate([ground, boxes, cables])
    nf = tris.shape[0] // 3
    m = default_materials(6)
    for k, col in enumerate([(0.5, 0.5, 0.5), (0.7, 0.3, 0.2), (0.2, 0.4, 0.7), (0.8, 0.8, 0.75), (0.3, 0.6, 0.3), (0.9, 0.85, 0.6)]):
        m["base_color"][k] = col
    m["metalness"][5] = 1.0
    mats = np.concatenate([[0, 0], 1 + (np.arange(12 * n_blocks) // 12) % 4, np.full(n_cables, 5)]).astype(np.uint32)
    assert mats.shape[0] == nf
    return _finish(tris, mats, m)


CITY_CAMERA = dict(origin=(0.0, 0.9, 2.4), fov=np.radians(50.0), F=100.0, focus=10000.0, forward=(0.0, -0.35, -1.0))
