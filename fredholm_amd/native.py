"""ctypes binding of libfredholm_hip.so (C ABI: include/fredholm_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FH_LIB") or os.path.join(_HERE, "libfredholm_hip.so")  # FH_LIB: developer override for A/B runs

FH_OK = 0
FLAG_TIME_KERNELS = 1
FLAG_COUNT_TRAVERSAL = 2
FLAG_SERIAL_PASSES = 8
FLAG_ROOT_START = 16  # (measurements) every ray starts its traversal at the root
FLAG_REFERENCE_FIRSTHIT = 4  # one fh_render(n_samples = k) = ONE reference launch of k samples, firsthit quirk included (pt.cu:432-433)

MATERIAL_DTYPE = np.dtype([
    ("diffuse", "f4"), ("base_color", "f4", 3), ("base_color_texture_id", "i4"), ("diffuse_roughness", "f4"),
    ("specular", "f4"), ("specular_color", "f4", 3), ("specular_color_texture_id", "i4"), ("specular_roughness", "f4"),
    ("specular_roughness_texture_id", "i4"),
    ("metalness", "f4"), ("metalness_texture_id", "i4"), ("metallic_roughness_texture_id", "i4"),
    ("coat", "f4"), ("coat_texture_id", "i4"), ("coat_color", "f4", 3), ("coat_roughness", "f4"), ("coat_roughness_texture_id", "i4"),
    ("transmission", "f4"), ("transmission_color", "f4", 3),
    ("sheen", "f4"), ("sheen_color", "f4", 3), ("sheen_roughness", "f4"),
    ("subsurface", "f4"), ("subsurface_color", "f4", 3),
    ("thin_walled", "f4"),
    ("emission", "f4"), ("emission_color", "f4", 3), ("emission_texture_id", "i4"),
    ("heightmap_texture_id", "i4"), ("normalmap_texture_id", "i4"), ("alpha_texture_id", "i4"),
])
assert MATERIAL_DTYPE.itemsize == 180


def default_materials(n):
    """n materials with the reference's defaults (fredholm/include/fredholm/shared.h:100-142)."""
    m = np.zeros(n, dtype=MATERIAL_DTYPE)
    m["diffuse"] = 1.0
    m["base_color"] = 1.0
    m["specular"] = 1.0
    m["specular_color"] = 1.0
    m["specular_roughness"] = 0.2
    m["coat_color"] = 1.0
    m["coat_roughness"] = 0.1
    m["transmission_color"] = 1.0
    m["sheen_color"] = 1.0
    m["sheen_roughness"] = 0.3
    m["subsurface_color"] = 1.0
    for k in MATERIAL_DTYPE.names:
        if k.endswith("texture_id"):
            m[k] = -1
    return m


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.c_void_p), ("srgb", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p),
                ("n_faces", C.c_uint32), ("indices", C.c_void_p), ("material_ids", C.c_void_p), ("instance_ids", C.c_void_p),
                ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_instances", C.c_uint32), ("object_to_world", C.c_void_p), ("world_to_object", C.c_void_p),
                ("n_textures", C.c_uint32), ("textures", C.c_void_p)]


class CameraC(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("fov", C.c_float), ("F", C.c_float), ("focus", C.c_float)]


class LayersC(C.Structure):
    _fields_ = [("beauty", C.c_void_p), ("position", C.c_void_p), ("depth", C.c_void_p), ("normal", C.c_void_p),
                ("texcoord", C.c_void_p), ("albedo", C.c_void_p)]


class PostParamsC(C.Structure):
    _fields_ = [("use_bloom", C.c_int32), ("bloom_threshold", C.c_float), ("bloom_sigma", C.c_float), ("ISO", C.c_float),
                ("chromatic_aberration", C.c_float)]


class StatsC(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("trace_closest_ms", C.c_double), ("trace_shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("n_closest_launches", C.c_uint64), ("n_shadow_launches", C.c_uint64),
                ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64), ("nodes_shadow", C.c_uint64), ("tris_shadow", C.c_uint64),
                ("paths", C.c_uint64), ("bvh_build_ms", C.c_double), ("bvh_nodes", C.c_uint64), ("bvh_node_bytes", C.c_uint64),
                ("bvh_tri_bytes", C.c_uint64),
                ("wave_node_steps_closest", C.c_uint64), ("wave_tri_steps_closest", C.c_uint64), ("wave_node_steps_shadow", C.c_uint64),
                ("wave_tri_steps_shadow", C.c_uint64), ("hist_nodes_closest", C.c_uint64 * 8), ("hist_nodes_shadow", C.c_uint64 * 8), ("tail_ms", C.c_double),
                ("generate_ms", C.c_double), ("accumulate_ms", C.c_double), ("queue_ms", C.c_double),
                ("n_generate_launches", C.c_uint64), ("n_accumulate_launches", C.c_uint64), ("n_shade_launches", C.c_uint64), ("n_tail_launches", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("bvh_depth", C.c_uint64), ("post_ms", C.c_double), ("n_post_launches", C.c_uint64),
                ("clk_cycles_closest", C.c_uint64), ("clk_ticks_closest", C.c_uint64), ("clk_cycles_shadow", C.c_uint64), ("clk_ticks_shadow", C.c_uint64),
                ("n_passes", C.c_uint64), ("sky_pixel_samples", C.c_uint64)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if hasattr(getattr(self, k), "__len__") else getattr(self, k)) for k, _ in self._fields_}


class FredholmError(RuntimeError):
    pass


# every symbol include/fredholm_hip.h and include/fredholm_hip_test.h (the known-answer hooks of the parity tests) declare (tests check the library exports all of them)
EXPORTS = [
    "fh_ctx_create", "fh_ctx_destroy", "fh_last_error", "fh_set_flags", "fh_get_flags", "fh_set_path_pool", "fh_path_pool_bytes", "fh_path_pool_allocated", "fh_alpha_face_counts", "fh_alpha_cell_counts", "fh_set_tail_depth", "fh_scene_upload", "fh_bvh_build", "fh_set_transforms",
    "fh_scene_n_lights", "fh_set_directional_light", "fh_clear_directional_light", "fh_set_sky_intensity", "fh_load_arhosek_sky",
    "fh_clear_arhosek_sky", "fh_load_ibl", "fh_clear_ibl", "fh_set_resolution", "fh_init_render_states", "fh_set_tile_shard", "fh_owned_pixel_count",
    "fh_pack_owned", "fh_unpack_shard", "fh_unpack_shards", "fh_render", "fh_sync", "fh_get_stats", "fh_reset_stats", "fh_post_process", "fh_denoise", "fh_gl_register_buffer", "fh_gl_unregister_buffer", "fh_malloc",
    "fh_free", "fh_memset", "fh_copy_to_device", "fh_copy_to_host", "fh_copy_on_device", "fh_image_load_rgba8", "fh_image_free", "fh_stream", "fh_trace_rays", "fh_kernel_info", "fh_kat_hash", "fh_kat_cmj",
    "fh_kat_sobol", "fh_kat_elementary", "fh_kat_warp", "fh_kat_bsdf", "fh_kat_bsdf_ior", "fh_kat_sky", "fh_kat_hosek_state", "fh_kat_camera",
    "fh_kat_offset_origin", "fh_kat_math", "fh_kat_sqrt", "fh_kat_tex2d", "fh_kat_face_classes", "fh_kat_alpha_records", "fh_kat_ray_start", "fh_measure_bandwidth",
]

_lib = None


def load_library(path=None):
    """Load libfredholm_hip.so; raises FredholmError when it has not been built (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FredholmError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(make -C fredholm_amd/csrc); the HIP path has no CPU fallback")
    try:
        L = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise FredholmError(f"cannot load {p}: {e}") from e
    L.fh_last_error.restype = C.c_char_p
    L.fh_last_error.argtypes = [C.c_void_p]
    L.fh_stream.restype = C.c_void_p
    L.fh_stream.argtypes = [C.c_void_p]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("fh_last_error", "fh_stream"):
            fn.restype = C.c_int
    _lib = L
    return L


def lib():
    return load_library()


def check(ctx, rc, what=""):
    if rc != FH_OK:
        msg = lib().fh_last_error(ctx)
        raise FredholmError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
