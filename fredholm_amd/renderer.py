"""Python mirror of the reference's host interface for the render path.

`Renderer`, `Camera`, `RenderLayer` and `PostProcessParams` keep the names, argument meaning and
call order of fredholm::Renderer (fredholm/include/fredholm/renderer.h:29-846), fredholm::Camera
(fredholm/include/fredholm/camera.h:22-135), RenderLayer (fredholm/include/fredholm/shared.h:201-208)
and PostProcessParams (fredholm/kernels/include/kernels/post-process.h:4-10); every method forwards
to the C ABI of libfredholm_hip.so.  Errors surface as FredholmError, the analogue of the
std::runtime_error the reference throws from CUDA_CHECK / OPTIX_CHECK.
"""
import ctypes as C
import os
import math

import numpy as np

from . import native as N


def look_at_transform(origin, forward=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0)):
    """camera-to-world 3x4 rows = inverse(lookAt(origin, origin + 0.01 forward, up)) (camera.h:51-69)."""
    o = np.asarray(origin, dtype=np.float64)
    f = np.asarray(forward, dtype=np.float64)
    f = f / np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r = r / np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.zeros((3, 4), dtype=np.float32)
    m[:, 0] = r
    m[:, 1] = u
    m[:, 2] = -f
    m[:, 3] = o
    return m


class Camera:
    """fredholm::Camera: m_transform (camera-to-world), m_fov (radians), m_F, m_focus (camera.h:22-69)."""

    def __init__(self, origin=(0.0, 0.0, 0.0), fov=0.5 * math.pi, F=8.0, focus=10000.0, forward=(0.0, 0.0, -1.0)):
        self.m_origin = tuple(float(v) for v in origin)
        self.m_forward = tuple(float(v) for v in forward)
        self.m_fov = float(fov)
        self.m_F = float(F)
        self.m_focus = float(focus)
        self.m_transform = look_at_transform(self.m_origin, self.m_forward)

    def set_origin(self, origin):
        self.m_origin = tuple(float(v) for v in origin)
        self.m_transform = look_at_transform(self.m_origin, self.m_forward)

    def params(self):
        """the 15 floats of CameraParams (shared.h:59-64)"""
        return np.concatenate([np.asarray(self.m_transform, dtype=np.float32).reshape(12), np.asarray([self.m_fov, self.m_F, self.m_focus], dtype=np.float32)])

    def as_c(self):
        c = N.CameraC()
        flat = np.asarray(self.m_transform, dtype=np.float32).reshape(12)
        for i in range(12):
            c.transform[i] = float(flat[i])
        c.fov, c.F, c.focus = self.m_fov, self.m_F, self.m_focus
        return c


class PostProcessParams:
    def __init__(self, use_bloom=False, bloom_threshold=2.0, bloom_sigma=5.0, ISO=80.0, chromatic_aberration=1.0):
        self.use_bloom, self.bloom_threshold, self.bloom_sigma, self.ISO, self.chromatic_aberration = use_bloom, bloom_threshold, bloom_sigma, ISO, chromatic_aberration

    def as_c(self):
        return N.PostParamsC(int(bool(self.use_bloom)), self.bloom_threshold, self.bloom_sigma, self.ISO, self.chromatic_aberration)


class DeviceBuffer:
    """Owning device allocation (role of cwl::CUDABuffer, cwl/include/cwl/buffer.h:18-85)."""

    def __init__(self, renderer, nbytes):
        self._r = renderer
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        N.check(renderer._ctx, N.lib().fh_malloc(renderer._ctx, C.c_uint64(self.nbytes), C.byref(p)), "fh_malloc")
        self.ptr = p.value

    def clear(self, value=0):
        N.check(self._r._ctx, N.lib().fh_memset(self._r._ctx, C.c_void_p(self.ptr), int(value), C.c_uint64(self.nbytes)), "fh_memset")

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        N.check(self._r._ctx, N.lib().fh_copy_to_device(self._r._ctx, C.c_void_p(self.ptr), N.ptr(a), C.c_uint64(a.nbytes)), "fh_copy_to_device")

    def download(self, dtype=np.float32, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        N.check(self._r._ctx, N.lib().fh_copy_to_host(self._r._ctx, N.ptr(out), C.c_void_p(self.ptr), C.c_uint64(self.nbytes)), "fh_copy_to_host")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr and self._r._ctx:
            N.lib().fh_free(self._r._ctx, C.c_void_p(self.ptr))
        self.ptr = None


class RenderLayer:
    """Six AOV buffers owned by the caller (shared.h:201-208; allocated by the apps, controller.cpp:80-107).

    Either allocated here through the C ABI, or wrapping external device pointers (e.g. torch tensors)."""

    NAMES = ("beauty", "position", "depth", "normal", "texcoord", "albedo")

    def __init__(self, renderer, width, height, pointers=None):
        self.width, self.height = int(width), int(height)
        self._bufs = {}
        self.ptrs = {}
        for name in self.NAMES:
            comps = 1 if name == "depth" else 4
            if pointers is not None:
                self.ptrs[name] = int(pointers[name])
            else:
                b = DeviceBuffer(renderer, self.width * self.height * comps * 4)
                b.clear()
                self._bufs[name] = b
                self.ptrs[name] = b.ptr

    def clear(self):
        for b in self._bufs.values():
            b.clear()

    def download(self, name):
        comps = 1 if name == "depth" else 4
        shape = (self.height, self.width) if comps == 1 else (self.height, self.width, 4)
        return self._bufs[name].download(np.float32, shape)

    def as_c(self):
        return N.LayersC(*(C.c_void_p(self.ptrs[n]) for n in self.NAMES))

    def free(self):
        for b in self._bufs.values():
            b.free()
        self._bufs = {}


class Renderer:
    """Drop-in for the method surface of fredholm::Renderer that the apps use on the render path.

    The OptiX pipeline-construction calls of the reference (create_module / create_program_group /
    create_pipeline / create_sbt, renderer.h:124-352) are accepted and ignored: there is no OptiX
    pipeline, the HIP kernels are compiled into the library."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        L = N.load_library()
        rc = L.fh_ctx_create(int(device), C.byref(self._ctx))
        if rc != N.FH_OK:
            msg = L.fh_last_error(None)
            self._ctx = None
            raise N.FredholmError(f"fh_ctx_create failed ({rc}): {msg.decode() if msg else ''}")
        self.m_width = self.m_height = 0
        self.seed = 1  # params.seed = 1 (renderer.h:664)
        self._keep = None
        self.m_scene = None

    # -- OptiX plumbing kept for call-compatibility
    def create_module(self, filepath=None):
        return None

    def create_program_group(self):
        return None

    def create_pipeline(self):
        return None

    def create_sbt(self):
        return None

    def close(self):
        if self._ctx:
            N.lib().fh_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        N.check(self._ctx, rc, what)

    def set_flags(self, flags):
        self._ck(N.lib().fh_set_flags(self._ctx, C.c_uint32(flags)), "fh_set_flags")

    def set_path_pool(self, target_paths):
        self._ck(N.lib().fh_set_path_pool(self._ctx, C.c_uint32(target_paths)), "fh_set_path_pool")

    def path_pool_bytes(self):
        """(bytes per path slot, pools) with the scene and lights as they are now: the pools take pools x target_paths x bytes of device memory"""
        b, n = C.c_uint64(0), C.c_uint32(0)
        self._ck(N.lib().fh_path_pool_bytes(self._ctx, C.byref(b), C.byref(n)), "fh_path_pool_bytes")
        return int(b.value), int(n.value)

    def alpha_face_counts(self):
        """(faces whose textures can cut, always pass, never pass, still tested) of the uploaded scene"""
        a = (C.c_uint32 * 4)()
        self._ck(N.lib().fh_alpha_face_counts(self._ctx, a), "fh_alpha_face_counts")
        return tuple(int(x) for x in a)

    def alpha_cell_counts(self):
        """(micromap cells of the faces that keep their any-hit test, always pass, never pass)"""
        a = (C.c_uint64 * 3)()
        self._ck(N.lib().fh_alpha_cell_counts(self._ctx, a), "fh_alpha_cell_counts")
        return tuple(int(x) for x in a)

    def path_pool_allocated(self):
        """(device bytes, path slots) the path pools hold right now, all pools together"""
        b, n = C.c_uint64(0), C.c_uint64(0)
        self._ck(N.lib().fh_path_pool_allocated(self._ctx, C.byref(b), C.byref(n)), "fh_path_pool_allocated")
        return int(b.value), int(n.value)

    def set_tail_depth(self, depth):
        self._ck(N.lib().fh_set_tail_depth(self._ctx, C.c_uint32(depth)), "fh_set_tail_depth")

    # -- scene (renderer.h:354-432)
    def load_scene(self, scene, clear=True):
        """scene: a .obj / .gltf path (renderer.h:354: load_scene(filepath, clear); clear=False appends, as rtcamp8.cpp:114-115 adds
        a camera .gltf to an .obj), a fredholm_amd.scene.Scene, or a dict of flat arrays as produced by fredholm_amd.scenes (the
        layout Scene exposes, scene.h:103-135)."""
        from .scene import Scene
        if isinstance(scene, (str, bytes)) or hasattr(scene, "__fspath__"):
            if self.m_scene is None:
                self.m_scene = Scene()
            self.m_scene.load_model(os.fspath(scene) if not isinstance(scene, bytes) else scene.decode(), clear)
            scene = self.m_scene.as_dict()
        elif isinstance(scene, Scene):
            self.m_scene = scene
            scene = scene.as_dict()
        else:
            self.m_scene = None  # flat arrays: no node hierarchy, no camera node
        v = np.ascontiguousarray(scene["vertices"], dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(scene["normals"], dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(scene["texcoords"], dtype=np.float32).reshape(-1, 2)
        idx = np.ascontiguousarray(scene["indices"], dtype=np.uint32).reshape(-1, 3)
        mid = np.ascontiguousarray(scene["material_ids"], dtype=np.uint32)
        mats = np.ascontiguousarray(scene["materials"])
        assert mats.dtype.itemsize == 180
        inst = scene.get("instance_ids")
        inst = None if inst is None else np.ascontiguousarray(inst, dtype=np.uint32)
        o2w = scene.get("object_to_world")
        w2o = scene.get("world_to_object")
        d = N.SceneDesc()
        d.n_vertices = v.shape[0]
        d.vertices, d.normals, d.texcoords = N.ptr(v), N.ptr(n), N.ptr(t)
        d.n_faces = idx.shape[0]
        d.indices, d.material_ids, d.instance_ids = N.ptr(idx), N.ptr(mid), N.ptr(inst)
        d.n_materials = mats.shape[0]
        d.materials = N.ptr(mats)
        if o2w is not None:
            o2w = np.ascontiguousarray(o2w, dtype=np.float32).reshape(-1, 12)
            w2o = np.ascontiguousarray(w2o, dtype=np.float32).reshape(-1, 12)
            d.n_instances = o2w.shape[0]
            d.object_to_world, d.world_to_object = N.ptr(o2w), N.ptr(w2o)
        textures = scene.get("textures") or []
        tex_arr = (N.TextureDesc * max(len(textures), 1))()
        tex_keep = []
        for k, tex in enumerate(textures):
            img = np.ascontiguousarray(tex["rgba8"], dtype=np.uint8)
            assert img.ndim == 3 and img.shape[2] == 4
            tex_keep.append(img)
            tex_arr[k] = N.TextureDesc(img.shape[1], img.shape[0], img.ctypes.data, int(bool(tex.get("srgb", False))))
        d.n_textures = len(textures)
        d.textures = C.cast(tex_arr, C.c_void_p) if textures else None
        self._keep = (v, n, t, idx, mid, mats, inst, o2w, w2o, tex_arr, tex_keep)
        self._ck(N.lib().fh_scene_upload(self._ctx, C.byref(d)), "fh_scene_upload")

    def build_gas(self):
        """renderer.h:434-496; the whole acceleration structure is built by build_ias' counterpart"""
        return None

    def build_ias(self):
        """renderer.h:498-552 -> on-device BVH build"""
        self._ck(N.lib().fh_bvh_build(self._ctx), "fh_bvh_build")

    def build_accel(self):
        self.build_ias()

    def set_transforms(self, object_to_world, world_to_object):
        o = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(-1, 12)
        w = np.ascontiguousarray(world_to_object, dtype=np.float32).reshape(-1, 12)
        self._ck(N.lib().fh_set_transforms(self._ctx, C.c_uint32(o.shape[0]), N.ptr(o), N.ptr(w)), "fh_set_transforms")

    def n_lights(self):
        out = C.c_uint32()
        self._ck(N.lib().fh_scene_n_lights(self._ctx, C.byref(out)), "fh_scene_n_lights")
        return out.value

    # -- environment (renderer.h:554-612)
    def set_directional_light(self, le, direction, angle):
        le = np.asarray(le, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        self._ck(N.lib().fh_set_directional_light(self._ctx, N.ptr(le), N.ptr(d), C.c_float(angle)), "fh_set_directional_light")

    def clear_directional_light(self):
        self._ck(N.lib().fh_clear_directional_light(self._ctx), "fh_clear_directional_light")

    def set_sky_intensity(self, v):
        self._ck(N.lib().fh_set_sky_intensity(self._ctx, C.c_float(v)), "fh_set_sky_intensity")

    def load_arhosek_sky(self, turbidity, albedo):
        self._ck(N.lib().fh_load_arhosek_sky(self._ctx, C.c_float(turbidity), C.c_float(albedo)), "fh_load_arhosek_sky")

    def clear_arhosek_sky(self):
        self._ck(N.lib().fh_clear_arhosek_sky(self._ctx), "fh_clear_arhosek_sky")

    def load_ibl(self, filepath_or_image):
        """renderer.h:574-581: a Radiance .hdr path (read like the reference's FloatTexture, scene.cpp:39-66) or the decoded
        lat-long image (H x W x 4 float32)"""
        if isinstance(filepath_or_image, (str, bytes)) or hasattr(filepath_or_image, "__fspath__"):
            from . import image_io
            filepath_or_image = image_io.load_hdr(filepath_or_image)
        img = np.ascontiguousarray(filepath_or_image, dtype=np.float32)
        assert img.ndim == 3 and img.shape[2] == 4
        self._ck(N.lib().fh_load_ibl(self._ctx, N.ptr(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0])), "fh_load_ibl")

    def clear_ibl(self):
        self._ck(N.lib().fh_clear_ibl(self._ctx), "fh_clear_ibl")

    # -- frame state (renderer.h:642-655)
    def set_resolution(self, width, height):
        self.m_width, self.m_height = int(width), int(height)
        self._ck(N.lib().fh_set_resolution(self._ctx, C.c_uint32(width), C.c_uint32(height)), "fh_set_resolution")

    def init_render_states(self):
        self._ck(N.lib().fh_init_render_states(self._ctx), "fh_init_render_states")

    def set_tile_shard(self, rank, world, tile_w=32, tile_h=32):
        self._ck(N.lib().fh_set_tile_shard(self._ctx, C.c_uint32(rank), C.c_uint32(world), C.c_uint32(tile_w), C.c_uint32(tile_h)), "fh_set_tile_shard")

    def owned_pixel_count(self):
        out = C.c_uint32()
        self._ck(N.lib().fh_owned_pixel_count(self._ctx, C.byref(out)), "fh_owned_pixel_count")
        return out.value

    def pack_owned(self, layer_ptr, floats_per_pixel, packed_ptr):
        self._ck(N.lib().fh_pack_owned(self._ctx, C.c_void_p(layer_ptr), C.c_uint32(floats_per_pixel), C.c_void_p(packed_ptr)), "fh_pack_owned")

    def unpack_shard(self, rank, world, packed_ptr, floats_per_pixel, layer_ptr):
        self._ck(N.lib().fh_unpack_shard(self._ctx, C.c_uint32(rank), C.c_uint32(world), C.c_void_p(packed_ptr), C.c_uint32(floats_per_pixel), C.c_void_p(layer_ptr)), "fh_unpack_shard")

    def unpack_shards(self, packed_ptrs, floats_per_pixel, layer_ptr):
        """fh_unpack_shards: every rank's packed shard (device pointers, rank order) into the frame in ONE launch"""
        arr = (C.c_void_p * len(packed_ptrs))(*[C.c_void_p(int(p)) for p in packed_ptrs])
        self._ck(N.lib().fh_unpack_shards(self._ctx, C.c_uint32(len(packed_ptrs)), arr, C.c_uint32(floats_per_pixel), C.c_void_p(layer_ptr)), "fh_unpack_shards")

    # -- the hot path (renderer.h:657-736)
    def set_time(self, time):
        """renderer.h:614-640: advance the animation, re-upload the instance transforms, rebuild the acceleration structure"""
        if self.m_scene is None:
            raise N.FredholmError("set_time: the scene was not loaded from a file or a Scene")
        self.m_scene.update_animation(time)
        o2w, w2o = self.m_scene.transforms_3x4()
        self.set_transforms(o2w, w2o)
        self.build_ias()

    def render(self, camera, bg_color, render_layer, n_samples, max_depth):
        cam = camera.as_c()
        if self.m_scene is not None and self.m_scene.m_has_camera_transform:  # renderer.h:670-676: a camera node overrides the pose
            flat = self.m_scene.camera_transform_3x4().reshape(12)
            for i in range(12):
                cam.transform[i] = float(flat[i])
        bg = np.asarray(bg_color, dtype=np.float32)
        layers = render_layer.as_c()
        self._ck(N.lib().fh_render(self._ctx, C.byref(cam), N.ptr(bg), C.byref(layers), C.c_uint32(n_samples), C.c_uint32(max_depth), C.c_uint32(self.seed)), "fh_render")

    def wait_for_completion(self):
        self._ck(N.lib().fh_sync(self._ctx), "fh_sync")

    def stats(self):
        s = N.StatsC()
        self._ck(N.lib().fh_get_stats(self._ctx, C.byref(s)), "fh_get_stats")
        return s.as_dict()

    def kernel_info(self, which):
        """fh_kernel_info: what the runtime reports for the streaming traversal kernel of the current scene (0: closest hit, 1: secondary rays)"""
        out = (C.c_uint32 * 6)()
        self._ck(N.lib().fh_kernel_info(self._ctx, int(which), out), "fh_kernel_info")
        return {"vgprs": int(out[0]), "static_lds_bytes": int(out[1]), "scratch_bytes": int(out[2]), "workgroups_per_cu": int(out[3]), "stack_levels_in_lds": int(out[4]),
                "stack_levels": int(out[5])}

    def shade_kernel_info(self):
        """fh_kernel_info(2 + c) for every shading class c of the scene: registers, LDS, scratch, waves per SIMD of the shade kernel each class runs"""
        out = []
        for c in range(8):
            o = (C.c_uint32 * 6)()
            if N.lib().fh_kernel_info(self._ctx, 2 + c, o) != 0:
                break
            out.append({"class": c, "lobes": int(o[5]), "compiled_for_lobes": int(o[4]), "vgprs": int(o[0]), "static_lds_bytes": int(o[1]), "scratch_bytes": int(o[2]), "waves_per_simd": int(o[3])})
        return out

    def reset_stats(self):
        self._ck(N.lib().fh_reset_stats(self._ctx), "fh_reset_stats")

    def stream(self):
        return N.lib().fh_stream(self._ctx)

    # -- post chain (kernels/post-process.h:126-128)
    def post_process(self, beauty_in_ptr, high_ptr, temp_ptr, width, height, params, beauty_out_ptr):
        pp = params.as_c()
        self._ck(N.lib().fh_post_process(self._ctx, C.c_void_p(beauty_in_ptr), C.c_void_p(high_ptr), C.c_void_p(temp_ptr), int(width), int(height), C.byref(pp), C.c_void_p(beauty_out_ptr)),
                 "fh_post_process")

    def denoise(self, width, height, beauty_ptr, normal_ptr, albedo_ptr, denoised_ptr, upscale=False):
        """the denoiser slot (Denoiser::denoise, denoiser.h:87-95): edge-avoiding a-trous filter guided by the normal and albedo layers"""
        self._ck(N.lib().fh_denoise(self._ctx, C.c_uint32(int(width)), C.c_uint32(int(height)), C.c_void_p(beauty_ptr), C.c_void_p(normal_ptr), C.c_void_p(albedo_ptr),
                                    C.c_void_p(denoised_ptr), int(bool(upscale))), "fh_denoise")

    # -- parity-test hooks
    def measure_bandwidth(self, nbytes=1 << 30, iters=8):
        """(read GB/s, copy GB/s) of this GPU's HBM, measured with streaming kernels (fh_measure_bandwidth)"""
        r, c = C.c_double(0.0), C.c_double(0.0)
        self._ck(N.lib().fh_measure_bandwidth(self._ctx, C.c_uint64(nbytes), C.c_uint32(iters), C.byref(r), C.byref(c)), "fh_measure_bandwidth")
        return r.value, c.value

    def trace_rays(self, rays7, any_hit=False):
        r = np.ascontiguousarray(rays7, dtype=np.float32).reshape(-1, 7)
        tuv = np.zeros((r.shape[0], 3), dtype=np.float32)
        prim = np.zeros(r.shape[0], dtype=np.uint32)
        self._ck(N.lib().fh_trace_rays(self._ctx, C.c_uint32(r.shape[0]), N.ptr(r), int(any_hit), N.ptr(tuv), N.ptr(prim)), "fh_trace_rays")
        return tuv, prim
