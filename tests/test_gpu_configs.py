"""GPU tests (-m gpu) at the full sizes of BASELINE.json configs[1], [3] and [4] (configs[2] lives in test_gpu_parity.py).

The CPU checker needs seconds per row at these sizes, so each test uses what is size-independent: determinism, batch (path-pool)
invariance, pixel-tile shard invariance, and bit-parity with the checker on a crop of rows of the full frame -- rows of the FULL-size
frame, not a smaller render: sampler keys and camera rays depend on the frame size.
"""
import ctypes as C

import numpy as np
import pytest

import fredholm_amd as F
from fredholm_amd import distributed as D
from fredholm_amd import scenes
from fredholm_amd.renderer import DeviceBuffer, PostProcessParams

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all())


def _rows_parity(gpu_rows, ref_rows):
    same = ((_bits(gpu_rows) == _bits(ref_rows)) | (np.isnan(gpu_rows) & np.isnan(ref_rows))).reshape(-1, gpu_rows.shape[-1] if gpu_rows.ndim == 3 else 1).all(axis=1).mean()
    assert same >= 0.999, f"only {same:.5f} of the pixels are bit-identical"


def _full_size_checks(oracle, sc, cam, w, h, depth, setup, rows, spp=2, bg=(0.0, 0.0, 0.0), shard=(3, 8)):
    """render spp samples of the full frame; re-render with one sample per pass; re-render as one rank of a tile split; compare the
    rows [rows[0], rows[1]) of all six AOVs with the checker.  Returns (renderer, layers, beauty) for further use."""
    r = F.Renderer(0)
    r.load_scene(sc)
    r.build_ias()
    setup(r)
    r.set_resolution(w, h)
    L = F.RenderLayer(r, w, h)
    r.render(cam, bg, L, spp, depth)
    r.wait_for_completion()
    a = {n: L.download(n) for n in F.RenderLayer.NAMES}
    assert (a["beauty"][..., 3] == 1).all() and a["beauty"][..., :3][np.isfinite(a["beauty"][..., :3])].mean() > 1e-3
    # one sample per pass, small pool: same bits (determinism + batching invariance at full size)
    L.clear()
    r.init_render_states()
    r.set_path_pool(w * h)
    for _ in range(spp):
        r.render(cam, bg, L, 1, depth)
    r.wait_for_completion()
    assert _same(a["beauty"], L.download("beauty"))
    # one rank of an interleaved tile split renders exactly its pixels and nothing else
    r.set_path_pool(1 << 25)
    r.set_tile_shard(shard[0], shard[1], 32, 32)
    L.clear()
    r.init_render_states()
    r.render(cam, bg, L, spp, depth)
    r.wait_for_completion()
    c = L.download("beauty").reshape(-1, 4)
    own = D.tile_ownership(w, h, shard[0], shard[1])
    assert _same(c[own], a["beauty"].reshape(-1, 4)[own])
    mask = np.ones(w * h, bool)
    mask[own] = False
    assert (c[mask] == 0).all()
    r.set_tile_shard(0, 1, 32, 32)
    # checker parity on a crop of rows
    S = oracle.Scene(sc)
    setup(S)
    Lo = S.new_layers(w, h)
    for _ in range(spp):
        S.render(cam.params(), w, h, Lo, 1, depth, bg=bg, n_threads=oracle.hardware_threads(), rows=rows)
    for name in F.RenderLayer.NAMES:
        _rows_parity(a[name][rows[0]:rows[1]], Lo[name][rows[0]:rows[1]])
    return r, L, a


def test_config1_cornell_area_light_1080p(oracle):
    cam = F.Camera(**scenes.CORNELL_CAMERA)
    r, L, a = _full_size_checks(oracle, scenes.cornell_box(), cam, 1920, 1080, 8, lambda x: None, rows=(400, 404))
    assert np.isfinite(a["beauty"]).all() and (a["depth"] > 0).mean() > 0.5  # the open front of the box fills the 16:9 frame's middle; every hit path is shaded up to 8 times
    r.close()


@pytest.fixture(scope="module")
def sponza(tmp_path_factory):
    from fredholm_amd import scenes_sponza as SS
    from fredholm_amd.scene import Scene
    path = tmp_path_factory.mktemp("sponza") / "sponza_like.gltf"
    info = SS.write_sponza_gltf(str(path))
    S = Scene()
    S.load_model(str(path))
    return SS, S, info, str(path)


def test_config3_sponza_class_gltf_1080p(oracle, sponza):
    """configs[3]: the Sponza-class textured glTF (277 k triangles, 26 PNG / JPEG textures, alpha cut-outs, metallic-roughness + normal
    maps, clearcoat, instanced node hierarchy) read from disk through the glTF loader, Hosek sky + sun, 1080p, depth 8"""
    SS, S, info, path = sponza
    sc = S.as_dict()
    assert sc["indices"].shape[0] == info["triangles"] >= 250_000 and len(sc["textures"]) >= 20
    cam = F.Camera(**SS.SPONZA_CAMERA)

    def setup(x):
        x.set_directional_light((12.0, 11.0, 9.0), SS.SPONZA_SUN, 1.0)
        x.load_arhosek_sky(3.0, 0.3)

    r, L, a = _full_size_checks(oracle, sc, cam, 1920, 1080, 8, setup, rows=(300, 303))
    # the frame shows what the asset is for: textured albedo (not constant), normal-mapped normals, cut-out foliage (alpha-rejected hits
    # let the rays through: the depth layer behind a plant is not the plant's)
    alb = a["albedo"][..., :3]
    assert alb.std() > 0.05 and (a["depth"] > 0).mean() > 0.7
    # loading the file path directly (Renderer.load_scene(path), renderer.h:354) gives the same frame as uploading the loader's arrays
    r2 = F.Renderer(0)
    r2.load_scene(path)
    r2.build_ias()
    setup(r2)
    r2.set_resolution(1920, 1080)
    L2 = F.RenderLayer(r2, 1920, 1080)
    r2.render(cam, (0.0, 0.0, 0.0), L2, 2, 8)
    r2.wait_for_completion()
    assert _same(L2.download("beauty"), a["beauty"])
    # any-hit agrees with closest-hit under alpha cut-outs on rays through the foliage
    rng = np.random.default_rng(4)
    n = 200000
    o = np.stack([rng.uniform(-2.8, 2.8, n), rng.uniform(0.05, 2.2, n), rng.uniform(-0.6, 0.6, n)], 1)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.full((n, 1), 1e9)], 1).astype(np.float32)
    tuv, prim = r.trace_rays(rays)
    occ = r.trace_rays(rays, any_hit=True)[1] != 0xFFFFFFFF
    assert np.array_equal(occ, prim != 0xFFFFFFFF)
    So = oracle.Scene(sc)
    tuv_o, prim_o = So.trace(rays[:20000])
    assert np.array_equal(prim[:20000], prim_o) and np.array_equal(_bits(tuv[:20000]), _bits(tuv_o))
    r.close()
    r2.close()


def test_config4_4k_depth16_with_post_chain(oracle):
    """configs[4]: 3840x2160, max_depth 16, emitters + Hosek sky, then bloom + chromatic aberration + tone map on the whole 4K frame"""
    sc = scenes.soup_with_emitters(1_000_000)
    cam = F.Camera(**scenes.SOUP_CAMERA)
    w, h = 3840, 2160

    def setup(x):
        x.set_directional_light((0.0, 0.0, 0.0), scenes.SOUP_SUN, 0.0)
        if isinstance(x, F.Renderer):
            x.clear_directional_light()
        else:
            oracle.lib().orc_set_directional_light(x.h, 0, None, None, C.c_float(0))
        x.load_arhosek_sky(3.0, 0.3)

    r, L, a = _full_size_checks(oracle, sc, cam, w, h, 16, setup, rows=(1079, 1081), shard=(5, 8))
    assert r.n_lights() == 2
    # post chain on the full 4K frame (the render above is in L again after the shard run: render the 2 spp once more, unsharded)
    L.clear()
    r.init_render_states()
    r.render(cam, (0.0, 0.0, 0.0), L, 2, 16)
    r.wait_for_completion()
    beauty = L.download("beauty")
    assert _same(beauty, a["beauty"])
    pp = PostProcessParams(use_bloom=True, bloom_threshold=2.0, bloom_sigma=5.0, ISO=80.0, chromatic_aberration=1.0)  # rtcamp8.cpp:57-60
    hi, tmp, out = (DeviceBuffer(r, w * h * 16) for _ in range(3))
    for b in (hi, tmp, out):
        b.clear()
    r.post_process(L.ptrs["beauty"], hi.ptr, tmp.ptr, w, h, pp, out.ptr)
    r.wait_for_completion()
    got = out.download(np.float32, (h, w, 4))
    assert np.isfinite(got).all() and got[..., :3].max() <= 1.0 and (got[..., 3][: h // 16 * 16] == 1).all()
    assert h % 16 == 0 and w % 16 == 0  # at 4K the floor-division grid (post-process.cu:9-11) covers the frame; at 1080p rows 1072..1079 stay unwritten:
    # the checker's post chain on a crop: bloom reads a 16-pixel halo, so rows [y0-16, y1+16) of the input decide rows [y0, y1) -- but the
    # chromatic-aberration fetch addresses the WHOLE frame (uv * width + width * (uv.y * height)), so the checker gets the full beauty layer
    # and only the compared rows are checked (the checker computes every row; ~1 minute on the GPU box's host cores is too slow, so it is
    # given a frame cropped in x instead: columns [0, 512) of every row, which keeps the row addressing intact)
    crop = np.ascontiguousarray(beauty[:, :512])
    r.set_resolution(512, h)
    hi2, tmp2, out2, in2 = (DeviceBuffer(r, 512 * h * 16) for _ in range(4))
    in2.upload(crop)
    for b in (hi2, tmp2, out2):
        b.clear()
    r.post_process(in2.ptr, hi2.ptr, tmp2.ptr, 512, h, pp, out2.ptr)
    r.wait_for_completion()
    got2 = out2.download(np.float32, (h, 512, 4))
    ref2 = oracle.post_process(crop, True, 2.0, 5.0, 80.0, 1.0)
    assert _same(got2, ref2)
    # 1080p leaves the last 8 rows untouched (post-process.cu:9-11: 1080 / 16 = 67 blocks)
    crop1080 = np.ascontiguousarray(beauty[:1080, :256])
    in3, hi3, tmp3, out3 = (DeviceBuffer(r, 256 * 1080 * 16) for _ in range(4))
    in3.upload(crop1080)
    for b in (hi3, tmp3, out3):
        b.clear()
    r.post_process(in3.ptr, hi3.ptr, tmp3.ptr, 256, 1080, pp, out3.ptr)
    r.wait_for_completion()
    got3 = out3.download(np.float32, (1080, 256, 4))
    assert (got3[1072:] == 0).all() and (got3[:1072, :, 3] == 1).all()
    assert _same(got3, oracle.post_process(crop1080, True, 2.0, 5.0, 80.0, 1.0))
    r.close()


def test_gl_interop_entry_points_fail_cleanly_without_an_opengl_context():
    """fh_gl_register_buffer (cwl::CUDAGLBuffer's backend) needs a current OpenGL context; on a headless box it must return an error, not take the
    process down.  Probed in a child process so that a misbehaving GL stack cannot end the test run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes as C, sys; sys.path.insert(0, %r)\n"
        "import fredholm_amd as F; from fredholm_amd import native as N\n"
        "r = F.Renderer(0); res, ptr, n = C.c_void_p(), C.c_void_p(), C.c_uint64()\n"
        "rc = N.lib().fh_gl_register_buffer(r._ctx, 12345, C.byref(res), C.byref(ptr), C.byref(n))\n"
        "print('rc', rc, N.lib().fh_last_error(r._ctx)); assert rc != 0 and not ptr.value\n"
        "assert N.lib().fh_gl_unregister_buffer(r._ctx, None) == 0\n"
        "r.close(); print('clean')\n") % root
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if run.returncode < 0:
        pytest.skip(f"the HIP runtime's GL interop died with signal {-run.returncode} without an OpenGL context on this box")
    assert run.returncode == 0 and "clean" in run.stdout, run.stdout + run.stderr
