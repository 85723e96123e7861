"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
fredholm_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_DATA = os.path.join(_HERE, "..", "fredholm_amd", "data")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "obsdf.h", "osampler.h", "ovec.h", "otexture.h")] + [os.path.join(_HERE, "..", "include", "fh_elementary.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        for n in ("orc_xxhash32_1", "orc_xxhash32_3", "orc_xxhash32_4", "orc_cmj_permute", "orc_sobol_raw", "orc_scene_n_lights"):
            getattr(L, n).restype = C.c_uint32
        L.orc_scene_create.restype = C.c_void_p
        rc = L.orc_init(os.path.normpath(_DATA).encode())
        if rc != 0:
            raise RuntimeError(f"oracle table load failed ({rc})")
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def xxhash32(*args):
    L = lib()
    a = [C.c_uint32(int(v) & 0xFFFFFFFF) for v in args]
    return {1: L.orc_xxhash32_1, 3: L.orc_xxhash32_3, 4: L.orc_xxhash32_4}[len(a)](*a)


def cmj_permute(i, l, p):
    return lib().orc_cmj_permute(C.c_uint32(i), C.c_uint32(l), C.c_uint32(p & 0xFFFFFFFF))


def cmj_2d(n_spp, scramble, depth, image_idx, count=1):
    out = np.zeros((count, 2), dtype=np.float32)
    lib().orc_cmj_2d(C.c_uint64(n_spp), C.c_uint32(scramble), C.c_uint32(depth), C.c_uint32(image_idx), int(count), _p(out))
    return out


def sobol_owen(index, dimension, seed, count=1):
    out = np.zeros(count, dtype=np.float32)
    lib().orc_sobol_owen(C.c_uint64(index), C.c_uint32(dimension), C.c_uint32(seed), int(count), _p(out))
    return out


def sobol_raw(index, dimension):
    return lib().orc_sobol_raw(C.c_uint64(index), C.c_uint32(dimension))


def offset_origin(p, n):
    out = np.zeros(3, dtype=np.float32)
    lib().orc_offset_origin(_p(np.asarray(p, dtype=np.float32)), _p(np.asarray(n, dtype=np.float32)), _p(out))
    return out


ELEMENTARY = {"sin": 0, "cos": 1, "exp": 2, "log": 3, "pow": 4, "acos": 5, "atan2": 6, "log2": 7, "pow1p5": 8}


def elementary(fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.zeros_like(x)
    lib().orc_elementary(ELEMENTARY[fn], int(x.size), _p(x), _p(y), _p(out))
    return out


def warp(kind, u, wo=None, alpha=None):
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 2)
    width = 3 if kind in (1, 3) else 2
    out = np.zeros((u.shape[0], width), dtype=np.float32)
    wo = None if wo is None else np.ascontiguousarray(wo, dtype=np.float32)
    alpha = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.float32)
    lib().orc_warp(int(kind), int(u.shape[0]), _p(u), _p(wo), _p(alpha), _p(out))
    return out


def bsdf(material, entering, wo, wi, u1, u2):
    m = np.ascontiguousarray(material)
    assert m.dtype.itemsize == 180
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    u1 = np.ascontiguousarray(u1, dtype=np.float32)
    u2 = np.ascontiguousarray(u2, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((wo.shape[0], 18), dtype=np.float32)
    lib().orc_bsdf(_p(m), int(bool(entering)), int(wo.shape[0]), _p(wo), _p(wi), _p(u1), _p(u2), _p(out))
    return out


def bsdf_ior(material, eta, wo, wi, u1, u2):
    """orc_bsdf with the relative index of refraction given (the reference's constructor fixes 1.5 / (1 / 1.5), bsdf.cu:16-18)"""
    m = np.ascontiguousarray(material)
    assert m.dtype.itemsize == 180
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    u1 = np.ascontiguousarray(u1, dtype=np.float32)
    u2 = np.ascontiguousarray(u2, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((wo.shape[0], 18), dtype=np.float32)
    lib().orc_bsdf_ior(_p(m), C.c_float(eta), int(wo.shape[0]), _p(wo), _p(wi), _p(u1), _p(u2), _p(out))
    return out


def hosek_cook(turbidity, albedo, sun_dir):
    out = np.zeros(30, dtype=np.float32)
    lib().orc_hosek_cook(C.c_float(turbidity), C.c_float(albedo), _p(np.asarray(sun_dir, dtype=np.float32)), _p(out))
    return out


def hosek_cook_elevation(turbidity, albedo, elevation):
    out = np.zeros(30, dtype=np.float32)
    lib().orc_hosek_cook_elevation(C.c_float(turbidity), C.c_float(albedo), C.c_float(elevation), _p(out))
    return out


_REF_HOSEK = None


def ref_hosek():
    """oracle/_ref/libref_hosek.so: the reference's own Hosek sources (arhosek.h, arhosek.cu) built by oracle/Makefile; None when it
    was not built (no /root/reference at build time)"""
    global _REF_HOSEK
    if _REF_HOSEK is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_hosek.so")
        if not os.path.exists(path):
            return None
        _REF_HOSEK = C.CDLL(path)
    return _REF_HOSEK


def ref_hosek_state(turbidity, albedo, elevation):
    """(configs[3][9], radiances[3]) from the reference's arhosek_rgb_skymodelstate_alloc_init (arhosek.h:298-322), run on the host"""
    cfg, rad = np.zeros(27, dtype=np.float32), np.zeros(3, dtype=np.float32)
    ref_hosek().ref_hosek_state(C.c_float(turbidity), C.c_float(albedo), C.c_float(elevation), _p(cfg), _p(rad))
    return cfg.reshape(3, 9), rad


def ref_hosek_radiance(turbidity, albedo, elevation, theta, gamma):
    """arhosek_tristim_skymodel_radiance (arhosek.cu:120-127) for every (theta, gamma), run on the GPU"""
    theta = np.ascontiguousarray(theta, dtype=np.float32)
    gamma = np.ascontiguousarray(gamma, dtype=np.float32)
    out = np.zeros((theta.shape[0], 3), dtype=np.float32)
    rc = ref_hosek().ref_hosek_radiance(C.c_float(turbidity), C.c_float(albedo), C.c_float(elevation), int(theta.shape[0]), _p(theta), _p(gamma), _p(out))
    if rc != 0:
        raise RuntimeError(f"ref_hosek_radiance: HIP error {rc}")
    return out


MATH_KINDS = {"albedo_reflection": (0, 3, 1), "albedo_sheen": (1, 2, 1), "onb": (2, 3, 6), "to_local": (3, 12, 3), "to_world": (4, 12, 3), "spherical": (5, 3, 2),
              "luminance": (6, 3, 1), "uchimura": (7, 3, 3), "linear_to_srgb": (8, 3, 3), "exposure": (9, 3, 2), "tone_map_tail": (10, 4, 3), "post_luminance": (11, 3, 1)}


def math(kind, x):
    """the checker's restatement of the small math blocks (kinds and widths as fh_kat_math)"""
    k, si, so = MATH_KINDS[kind]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, si)
    out = np.zeros((x.shape[0], so), dtype=np.float32)
    lib().orc_math(k, int(x.shape[0]), _p(x), _p(out))
    return out


_REF_LMP = None


def ref_lut_math_post():
    """oracle/_ref/libref_lut_math_post.so: the reference's own lut.cu, math.cu and kernels/post-process.h built for the host by oracle/Makefile;
    None when it was not built (no /root/reference at build time)"""
    global _REF_LMP
    if _REF_LMP is None:
        path = os.path.join(_HERE, "_ref", "libref_lut_math_post.so")
        if not os.path.exists(path):
            return None
        _REF_LMP = C.CDLL(path)
    return _REF_LMP


def ref_math(kind, x):
    """the REFERENCE's own functions for the same kinds, run on the host (glibc libm where they call powf / expf / acosf ...)"""
    R = ref_lut_math_post()
    k, si, so = MATH_KINDS[kind]
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, si)
    n = int(x.shape[0])
    out = np.zeros((n, so), dtype=np.float32)
    col = lambda j: np.ascontiguousarray(x[:, j])
    blk = lambda j: np.ascontiguousarray(x[:, 3 * j:3 * j + 3])
    if kind == "albedo_reflection":
        a, b, c = col(0), col(1), col(2); R.ref_albedo_reflection(n, _p(a), _p(b), _p(c), _p(out))
    elif kind == "albedo_sheen":
        a, b = col(0), col(1); R.ref_albedo_sheen(n, _p(a), _p(b), _p(out))
    elif kind == "onb":
        t, b = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        R.ref_orthonormal_basis(n, _p(x), _p(t), _p(b)); out = np.concatenate([t, b], axis=1)
    elif kind in ("to_local", "to_world"):
        v, t, nn, b = blk(0), blk(1), blk(2), blk(3)
        getattr(R, "ref_world_to_local" if kind == "to_local" else "ref_local_to_world")(n, _p(v), _p(t), _p(nn), _p(b), _p(out))
    elif kind == "spherical":
        R.ref_cartesian_to_spherical(n, _p(x), _p(out))
    elif kind == "luminance":
        R.ref_rgb_to_luminance(n, _p(x), _p(out))
    elif kind == "post_luminance":
        R.ref_post_luminance(n, _p(x), _p(out))
    elif kind == "uchimura":
        R.ref_uchimura(n, _p(x), _p(out))
    elif kind == "linear_to_srgb":
        R.ref_linear_to_srgb(n, _p(x), _p(out))
    elif kind == "exposure":
        a, b, c = col(0), col(1), col(2)
        ev, ex = np.zeros(n, np.float32), np.zeros(n, np.float32)
        R.ref_exposure(n, _p(a), _p(b), _p(c), _p(ev), _p(ex)); out = np.stack([ev, ex], axis=1)
    elif kind == "tone_map_tail":
        iso = np.unique(x[:, 3])
        for v in iso:  # the reference's kernel takes one ISO per launch
            m = x[:, 3] == v
            rgb = np.ascontiguousarray(x[m, :3]); o = np.zeros_like(rgb)
            R.ref_tone_map_tail(int(rgb.shape[0]), C.c_float(float(v)), _p(rgb), _p(o)); out[m] = o
    else:
        raise KeyError(kind)
    return out


def ref_albedo_reflection_ior1(x):
    """the reference's compute_directional_albedo_reflection_ior1 (lut.cu:1038-1045) for rows (w.y, roughness, eta)"""
    R = ref_lut_math_post()
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3)
    a, b, c = (np.ascontiguousarray(x[:, j]) for j in range(3))
    out = np.zeros(x.shape[0], dtype=np.float32)
    R.ref_albedo_reflection_ior1(int(x.shape[0]), _p(a), _p(b), _p(c), _p(out))
    return out


def ref_lut_tables():
    R = ref_lut_math_post()
    a, b = np.zeros(512, np.float32), np.zeros(256, np.float32)
    R.ref_lut_entries(_p(a), _p(b))
    return a, b


def hosek_radiance(state30, sun_dir, intensity, dirs):
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(d)
    lib().orc_hosek_radiance(_p(np.ascontiguousarray(state30, dtype=np.float32)), _p(np.asarray(sun_dir, dtype=np.float32)), C.c_float(intensity), int(d.shape[0]), _p(d), _p(out))
    return out


def camera_rays(cam15, width, height, seed, pixel_idx, n_spp):
    pix = np.ascontiguousarray(pixel_idx, dtype=np.uint32)
    ns = np.ascontiguousarray(n_spp, dtype=np.uint32)
    out = np.zeros((pix.size, 6), dtype=np.float32)
    lib().orc_camera_rays(_p(np.ascontiguousarray(cam15, dtype=np.float32)), C.c_uint32(width), C.c_uint32(height), C.c_uint32(seed), int(pix.size), _p(pix), _p(ns), _p(out))
    return out


def post_process(img, use_bloom, threshold, sigma, iso, ca):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    hi = np.zeros_like(img)
    tmp = np.zeros_like(img)
    out = np.zeros_like(img)
    lib().orc_post_process(_p(img), _p(hi), _p(tmp), int(w), int(h), int(bool(use_bloom)), C.c_float(threshold), C.c_float(sigma), C.c_float(iso), C.c_float(ca), _p(out))
    return out


def tex2d(rgba8, srgb, uv):
    """sample an 8-bit RGBA texture (H x W x 4) with the normative texture unit"""
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((uv.shape[0], 4), dtype=np.float32)
    lib().orc_tex2d(_p(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0]), int(bool(srgb)), int(uv.shape[0]), _p(uv), _p(out))
    return out


def denoise(beauty, normal, albedo, upscale=False):
    """the checker's restatement of the denoiser slot (edge-avoiding a-trous filter; fh_denoise)"""
    b, n, a = (np.ascontiguousarray(x, dtype=np.float32) for x in (beauty, normal, albedo))
    h, w = b.shape[:2]
    out = np.zeros((2 * h, 2 * w, 4) if upscale else (h, w, 4), np.float32)
    lib().orc_denoise(C.c_uint32(w), C.c_uint32(h), _p(b), _p(n), _p(a), _p(out), int(bool(upscale)))
    return out


def tex2d_f32(rgba32f, uv):
    """the same for a float4 texture (the lat-long IBL path)"""
    img = np.ascontiguousarray(rgba32f, dtype=np.float32)
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((uv.shape[0], 4), dtype=np.float32)
    lib().orc_tex2d_f32(_p(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0]), int(uv.shape[0]), _p(uv), _p(out))
    return out


def hardware_threads():
    return lib().orc_hardware_threads()


class Scene:
    """CPU scene + BVH of the checker; mirrors the arrays Renderer.load_scene takes."""

    def __init__(self, scene):
        L = lib()
        v = np.ascontiguousarray(scene["vertices"], dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(scene["normals"], dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(scene["texcoords"], dtype=np.float32).reshape(-1, 2)
        idx = np.ascontiguousarray(scene["indices"], dtype=np.uint32).reshape(-1, 3)
        mid = np.ascontiguousarray(scene["material_ids"], dtype=np.uint32)
        mats = np.ascontiguousarray(scene["materials"])
        inst = scene.get("instance_ids")
        inst = None if inst is None else np.ascontiguousarray(inst, dtype=np.uint32)
        o2w, w2o = scene.get("object_to_world"), scene.get("world_to_object")
        nxf = 0
        if o2w is not None:
            o2w = np.ascontiguousarray(o2w, dtype=np.float32).reshape(-1, 12)
            w2o = np.ascontiguousarray(w2o, dtype=np.float32).reshape(-1, 12)
            nxf = o2w.shape[0]
        class _TexDesc(C.Structure):
            _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p), ("srgb", C.c_int32)]

        textures = scene.get("textures") or []
        arr = (_TexDesc * max(len(textures), 1))()
        keep = []
        for k, tex in enumerate(textures):
            img = np.ascontiguousarray(tex["rgba8"], dtype=np.uint8)
            keep.append(img)
            arr[k] = _TexDesc(img.shape[1], img.shape[0], img.ctypes.data, int(bool(tex.get("srgb", False))))
        self.h = L.orc_scene_create(C.c_uint32(v.shape[0]), _p(v), _p(n), _p(t), C.c_uint32(idx.shape[0]), _p(idx), _p(mid), _p(inst), C.c_uint32(mats.shape[0]), _p(mats),
                                    C.c_uint32(nxf), _p(o2w), _p(w2o), C.c_uint32(len(textures)), C.cast(arr, C.c_void_p))
        if not self.h:
            raise RuntimeError("oracle: scene rejected (material references a texture id outside the texture list)")
        self.h = C.c_void_p(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def n_lights(self):
        return lib().orc_scene_n_lights(self.h)

    def set_directional_light(self, le, direction, angle):
        lib().orc_set_directional_light(self.h, 1, _p(np.asarray(le, dtype=np.float32)), _p(np.asarray(direction, dtype=np.float32)), C.c_float(angle))

    def set_sky_intensity(self, v):
        lib().orc_set_sky_intensity(self.h, C.c_float(v))

    def load_ibl(self, rgba32f):
        img = np.ascontiguousarray(rgba32f, dtype=np.float32)
        lib().orc_set_ibl(self.h, _p(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0]))

    def clear_ibl(self):
        lib().orc_set_ibl(self.h, None, 0, 0)

    def load_arhosek_sky(self, turbidity, albedo):
        lib().orc_set_hosek(self.h, 1, C.c_float(turbidity), C.c_float(albedo))

    def trace(self, rays7, any_hit=False, brute=False):
        r = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        tuv = np.zeros((r.shape[0], 3), dtype=np.float32)
        prim = np.zeros(r.shape[0], dtype=np.uint32)
        lib().orc_trace(self.h, int(r.shape[0]), _p(r), int(any_hit), int(brute), _p(tuv), _p(prim))
        return tuv, prim

    def new_layers(self, width, height):
        return {"beauty": np.zeros((height, width, 4), np.float32), "position": np.zeros((height, width, 4), np.float32), "depth": np.zeros((height, width), np.float32),
                "normal": np.zeros((height, width, 4), np.float32), "texcoord": np.zeros((height, width, 4), np.float32), "albedo": np.zeros((height, width, 4), np.float32),
                "sample_count": np.zeros((height, width), np.uint32)}

    def render(self, cam15, width, height, layers, n_samples, max_depth, bg=(0, 0, 0), seed=1, n_threads=1, rows=None):
        y0, y1 = (0, height) if rows is None else rows
        lib().orc_render(self.h, _p(np.ascontiguousarray(cam15, dtype=np.float32)), C.c_uint32(width), C.c_uint32(height), _p(np.asarray(bg, dtype=np.float32)), C.c_uint32(seed),
                         C.c_uint32(n_samples), C.c_uint32(max_depth), _p(layers["beauty"]), _p(layers["position"]), _p(layers["depth"]), _p(layers["normal"]), _p(layers["texcoord"]),
                         _p(layers["albedo"]), _p(layers["sample_count"]), int(n_threads), C.c_uint32(y0), C.c_uint32(y1))
        return layers
