/* fredholm_hip.h -- C ABI of libfredholm_hip.so, the MI355X (gfx950) replacement for the render
 * loop of yumcyaWiz/fredholm.
 *
 * The reference has no FFI: applications call the header-only C++ class fredholm::Renderer
 * (fredholm/include/fredholm/renderer.h) directly.  Each entry point below is what one of its
 * methods forwards to in the drop-in facade (include/fredholm/renderer.h in this repo); the
 * reference method it replaces is cited as file:line.  All arguments are plain pointers and
 * sizes; device pointers are HIP device addresses owned by the caller unless stated otherwise.
 *
 * Error convention: every function returns FH_OK (0) or a negative FH_E_* code and records a
 * message retrievable with fh_last_error(); the C++ facade turns non-zero into
 * std::runtime_error, matching the reference's CUDA_CHECK / OPTIX_CHECK behaviour
 * (cwl/include/cwl/util.h:11-34, optwl/include/optwl/optwl.h:11-35).
 * Threading: one host thread drives a context (reference: gui thread / rtcamp8 render thread);
 * fh_render is asynchronous on the context's private HIP stream, fh_sync blocks.
 */
#ifndef FREDHOLM_HIP_H
#define FREDHOLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FH_OK 0
#define FH_E_INVALID -1      /* bad argument / state */
#define FH_E_HIP -2          /* a HIP runtime call failed */
#define FH_E_UNSUPPORTED -3  /* feature present in the reference but not in this build (glTF, image file decoding) */

typedef struct fh_ctx fh_ctx;

/* 180-byte material record, field-for-field the reference's Material (fredholm/include/fredholm/shared.h:100-142) */
typedef struct fh_material {
  float diffuse; float base_color[3]; int32_t base_color_texture_id; float diffuse_roughness;
  float specular; float specular_color[3]; int32_t specular_color_texture_id; float specular_roughness; int32_t specular_roughness_texture_id;
  float metalness; int32_t metalness_texture_id; int32_t metallic_roughness_texture_id;
  float coat; int32_t coat_texture_id; float coat_color[3]; float coat_roughness; int32_t coat_roughness_texture_id;
  float transmission; float transmission_color[3];
  float sheen; float sheen_color[3]; float sheen_roughness;
  float subsurface; float subsurface_color[3];
  float thin_walled;
  float emission; float emission_color[3]; int32_t emission_texture_id;
  int32_t heightmap_texture_id; int32_t normalmap_texture_id; int32_t alpha_texture_id;
} fh_material;

/* 8-bit RGBA texture as the reference's Texture holds it after stb_image load (fredholm/src/scene.cpp:7-37: 4 channels,
 * v-flipped on load) plus its TextureType: srgb = 1 for COLOR textures (hardware sRGB decode, cwl/texture.h:35-47) */
typedef struct fh_texture_desc {
  uint32_t width, height;
  const uint8_t* rgba8;
  int32_t srgb;
} fh_texture_desc;

/* Host-side flat scene, exactly the arrays Renderer::load_scene uploads (renderer.h:361-421; Scene members scene.h:103-135).
 * transforms are 3x4 row-major object_to_world / world_to_object per instance (renderer.h:404-421); NULL / 0 = one identity. */
typedef struct fh_scene_desc {
  uint32_t n_vertices;
  const float* vertices;  /* float3[n_vertices] */
  const float* normals;   /* float3[n_vertices] */
  const float* texcoords; /* float2[n_vertices] */
  uint32_t n_faces;
  const uint32_t* indices;      /* uint3[n_faces] */
  const uint32_t* material_ids; /* uint[n_faces]  */
  const uint32_t* instance_ids; /* uint[n_faces], may be NULL (all 0) */
  uint32_t n_materials;
  const fh_material* materials;
  uint32_t n_instances;
  const float* object_to_world; /* float[12] per instance, may be NULL */
  const float* world_to_object; /* float[12] per instance, may be NULL */
  uint32_t n_textures;           /* textures referenced by the materials' *_texture_id fields (renderer.h:372-386) */
  const fh_texture_desc* textures;
} fh_scene_desc;

/* CameraParams (shared.h:59-64): camera-to-world 3x4 rows, vertical fov in radians, F-number, focus distance */
typedef struct fh_camera {
  float transform[12];
  float fov, F, focus;
} fh_camera;

/* RenderLayer (shared.h:201-208): six caller-owned device buffers of width*height elements */
typedef struct fh_render_layers {
  float* beauty;   /* float4 */
  float* position; /* float4 */
  float* depth;    /* float  */
  float* normal;   /* float4 */
  float* texcoord; /* float4 */
  float* albedo;   /* float4 */
} fh_render_layers;

/* PostProcessParams (fredholm/kernels/include/kernels/post-process.h:4-10) */
typedef struct fh_post_params {
  int32_t use_bloom;
  float bloom_threshold, bloom_sigma, ISO, chromatic_aberration;
} fh_post_params;

/* counters / timers of the last fh_render .. fh_sync interval, see fh_get_stats */
typedef struct fh_stats {
  double render_ms;        /* HIP-event time of the whole fh_render submission on the context stream */
  double trace_closest_ms; /* summed HIP-event time of the closest-hit traversal kernel launches */
  double trace_shadow_ms;  /* summed time of the any-hit / secondary traversal kernel launches */
  double shade_ms;         /* summed time of the shade kernels */
  uint64_t n_closest_launches, n_shadow_launches;
  uint64_t rays_closest, rays_shadow;          /* rays traced by the two traversal kernels (only when FH_FLAG_COUNT_TRAVERSAL) */
  uint64_t nodes_closest, tris_closest;        /* node visits / triangle tests (only when FH_FLAG_COUNT_TRAVERSAL) */
  uint64_t nodes_shadow, tris_shadow;
  uint64_t paths;                              /* camera paths started */
  double bvh_build_ms;
  uint64_t bvh_nodes, bvh_node_bytes, bvh_tri_bytes;
  /* instrumented build only: wave-level executions of the node test / triangle test (SIMD efficiency =
   * nodes_* / (64 * wave_node_steps_*), likewise for triangles) */
  uint64_t wave_node_steps_closest, wave_tri_steps_closest, wave_node_steps_shadow, wave_tri_steps_shadow;
  /* instrumented build only: rays by number of node visits: <= 8, 16, 32, 64, 128, 256, 512, more */
  uint64_t hist_nodes_closest[8], hist_nodes_shadow[8];
  double tail_ms; /* summed HIP-event time of the k_tail launches */
  double generate_ms, accumulate_ms, queue_ms; /* k_generate + k_bump_issued ; k_accumulate ; k_route + the cell sorts of the bounce queues */
  uint64_t n_generate_launches, n_accumulate_launches, n_shade_launches, n_tail_launches;
  uint64_t shaded_hits; /* surface hits shaded by the k_shade kernels (only when FH_FLAG_COUNT_TRAVERSAL) */
  uint64_t bvh_depth;   /* node levels of the wide BVH; a traversal stack needs levels - 1 entries, of which the streaming kernels may keep only the first in LDS (fh_kernel_info) */
  double post_ms;       /* summed HIP-event time of the fh_post_process chains (threshold + blur + tone map) */
  uint64_t n_post_launches;
  /* FH_FLAG_TIME_KERNELS, streaming traversal kernels: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) summed over their waves;
   * the clock the chip held while they ran = cycles / ticks x 100 MHz */
  uint64_t clk_cycles_closest, clk_ticks_closest, clk_cycles_shadow, clk_ticks_shadow;
  uint64_t n_passes;          /* passes of the path pools submitted */
  uint64_t sky_pixel_samples; /* camera samples (counted in `paths`) rendered by the sky-pixel kernel: samples of pixels no ray of which can reach the scene's bounds */
} fh_stats;

#define FH_FLAG_TIME_KERNELS 1u    /* bracket traversal/shade launches with HIP events (fh_stats *_ms) */
#define FH_FLAG_COUNT_TRAVERSAL 2u /* instrumented traversal kernels: count node visits / triangle tests */
/* bug-compatible multi-sample launches.  fh_render(n_samples = k) is by default k one-sample launches.  With this flag it reproduces
 * what ONE reference launch of k samples computes: the reference never resets payload.firsthit inside a launch (pt.cu:432-433), so only
 * the first sample of the launch that hits anything records AOVs / sees emitters directly (:745-760), primary misses after it add no sky
 * (:509) and every sample averages that first hit's AOVs again (:483-487).  rtcamp8 renders 16 samples per launch (rtcamp8.cpp:183-189). */
#define FH_FLAG_REFERENCE_FIRSTHIT 4u
/* run the passes of fh_render one after the other on the main stream instead of two in flight: slower, but every kernel then runs alone
 * on the GPU, so the HIP-event times of FH_FLAG_TIME_KERNELS are kernel times (with passes in flight they include the other stream's work) */
#define FH_FLAG_SERIAL_PASSES 8u
/* measurements: every ray starts its traversal at the root, whatever the library decided for the scene (render.hip: where a pass's first-hit rays start); results do not
 * depend on it -- bench.py counts with it what a walk from the root would have tested */
#define FH_FLAG_ROOT_START 16u

/* -- context: replaces optwl::Context + Renderer ctor/dtor (optwl.h:41-81, renderer.h:32-122) */
int fh_ctx_create(int device, fh_ctx** out);
int fh_ctx_destroy(fh_ctx* ctx);
const char* fh_last_error(fh_ctx* ctx); /* ctx may be NULL for creation errors */
int fh_set_flags(fh_ctx* ctx, uint32_t flags);
int fh_get_flags(fh_ctx* ctx, uint32_t* flags); /* (a caller that wants to change one flag reads, edits and sets) */
/* target number of camera paths in flight per pass (path-pool slots); a pass starts floor(target / owned pixels) >= 1
 * samples per pixel.  Results do not depend on it.  Default 32 Mi paths per pool, three pools (one per pass in flight), 284-436 bytes per path.  Whenever a
 * default-sized pool has to be allocated (first frame, after fh_scene_upload changed what a path record holds) the default is lowered so that all pools together
 * stay within a quarter of the device memory that is free at that moment (memory the pools already hold counts as free); a size set here is taken as given.
 * The target is an upper bound: a pool is allocated for the paths its passes really start -- (owned pixels that can see the scene) x (samples per pass) -- and only grows. */
int fh_set_path_pool(fh_ctx* ctx, uint32_t target_paths);
/* device memory of the path pools with the scene and lights as they are now: bytes per path slot and the number of pools (one per pass in flight);
 * a caller that sizes the pools for a frame (bench.py) multiplies: pools x target_paths x bytes_per_path */
int fh_path_pool_bytes(fh_ctx* ctx, uint64_t* bytes_per_path, uint32_t* pools);
/* cut-out faces of the uploaded scene: [0] faces whose textures can discard a hit (pt.cu:545-678), [1] of them: the any-hit test passes wherever the face can be hit (no test
 * at run time), [2] it never passes (no ray can hit the face), [3] faces that keep their test.  Decided per face at fh_scene_upload from the texels the face can address. */
int fh_alpha_face_counts(fh_ctx* ctx, uint32_t counts[4]);
/* the opacity micromaps of the faces that keep their test (16 x 16 cells of a face's barycentrics, decided at upload like the faces themselves): [0] cells, [1] of them: the
 * test passes everywhere in the cell, [2] nowhere; candidates in such cells are decided by a two-bit look-up */
int fh_alpha_cell_counts(fh_ctx* ctx, uint64_t counts[3]);
/* what the path pools hold right now: device bytes of all pools together and path slots (summed over the pools) */
int fh_path_pool_allocated(fh_ctx* ctx, uint64_t* bytes, uint64_t* paths);
/* number of bounces run as bounce-synchronous wavefront kernels before the surviving paths are finished by one
 * fused launch (k_tail).  Results do not depend on it.  0 (default) = adaptive: the depth at which fewer than 64 Ki
 * paths survived in earlier passes; a value >= max_depth disables the fused tail. */
int fh_set_tail_depth(fh_ctx* ctx, uint32_t depth);

/* -- scene: Renderer::load_scene upload + AreaLight extraction (renderer.h:354-432) */
int fh_scene_upload(fh_ctx* ctx, const fh_scene_desc* scene);
/* Renderer::build_gas + build_ias (renderer.h:434-552): on-device LBVH build + wide-BVH collapse */
int fh_bvh_build(fh_ctx* ctx);
/* Renderer::set_time's transform re-upload + IAS rebuild (renderer.h:614-640); call fh_bvh_build afterwards */
int fh_set_transforms(fh_ctx* ctx, uint32_t n_instances, const float* object_to_world, const float* world_to_object);
int fh_scene_n_lights(fh_ctx* ctx, uint32_t* out);

/* -- environment: renderer.h:554-612.  le/dir are float[3]; angle in degrees. */
int fh_set_directional_light(fh_ctx* ctx, const float* le, const float* dir, float angle);
int fh_clear_directional_light(fh_ctx* ctx);
int fh_set_sky_intensity(fh_ctx* ctx, float intensity);
int fh_load_arhosek_sky(fh_ctx* ctx, float turbidity, float albedo); /* renderer.h:588-607 */
int fh_clear_arhosek_sky(fh_ctx* ctx);                               /* renderer.h:609-612 */
int fh_load_ibl(fh_ctx* ctx, const float* rgba, uint32_t w, uint32_t h); /* renderer.h:574-581: float4 lat-long image, already decoded */
int fh_clear_ibl(fh_ctx* ctx);                                          /* renderer.h:583-586 */

/* -- frame state: renderer.h:642-655 */
int fh_set_resolution(fh_ctx* ctx, uint32_t width, uint32_t height); /* also resets the sample counters */
int fh_init_render_states(fh_ctx* ctx);                              /* sample_count = 0 */
/* pixel-tile sharding for multi-GPU rendering: this context renders only the tiles t with t % world == rank
 * (tiles of tile_w x tile_h pixels, row-major tile order).  world = 1 renders everything (default). */
int fh_set_tile_shard(fh_ctx* ctx, uint32_t rank, uint32_t world, uint32_t tile_w, uint32_t tile_h);
/* number of pixels this context owns, and pack/unpack of owned pixels for the framebuffer gather */
int fh_owned_pixel_count(fh_ctx* ctx, uint32_t* out);
int fh_pack_owned(fh_ctx* ctx, const float* layer, uint32_t floats_per_pixel, float* packed);
int fh_unpack_shard(fh_ctx* ctx, uint32_t rank, uint32_t world, const float* packed, uint32_t floats_per_pixel, float* layer);
/* the same for ALL ranks of a split in one asynchronous launch: packed[r] = rank r's packed shard (device pointers in a host array of `world` entries, read during the call).
 * What rank 0 calls once per presented frame after the gather (bench.py); world > 16 falls back to one launch per rank. */
int fh_unpack_shards(fh_ctx* ctx, uint32_t world, const float* const* packed, uint32_t floats_per_pixel, float* layer);

/* -- THE hot path: Renderer::render (renderer.h:657-734) -> __raygen__rg & friends (fredholm/modules/pt.cu:418-999).
 * Adds n_samples samples per owned pixel to the running means in `layers`; equivalent to n_samples consecutive
 * reference launches with n_samples = 1 (the reference's only well-defined mode). seed: reference uses 1. */
int fh_render(fh_ctx* ctx, const fh_camera* camera, const float* bg_color, const fh_render_layers* layers, uint32_t n_samples, uint32_t max_depth, uint32_t seed);
int fh_sync(fh_ctx* ctx); /* Renderer::wait_for_completion (renderer.h:736) */
int fh_get_stats(fh_ctx* ctx, fh_stats* out);
int fh_reset_stats(fh_ctx* ctx);

/* -- post chain: post_process_kernel_launch (fredholm/kernels/src/post-process.cu:5-35); all device pointers, float4 images */
int fh_post_process(fh_ctx* ctx, const float* beauty_in, float* beauty_high_luminance, float* beauty_temp, int width, int height, const fh_post_params* params, float* beauty_out);

/* Denoiser slot (Denoiser::denoise, fredholm/include/fredholm/denoiser.h:87-95).  The reference invokes NVIDIA's OptiX AI denoiser (HDR model with
 * albedo + normal guide layers, optionally the 2x upscaling model), a proprietary network; the slot is filled by an edge-avoiding a-trous wavelet
 * filter (Dammertz et al. 2010) on albedo-demodulated radiance with the same inputs and output: float4 beauty / normal / albedo layers of
 * width x height pixels in, float4 denoised out (2*width x 2*height when upscale2x, by pixel replication).  Asynchronous on the context stream. */
int fh_denoise(fh_ctx* ctx, uint32_t width, uint32_t height, const float* beauty, const float* normal, const float* albedo, float* denoised, int upscale2x);

/* OpenGL interop for display (cwl::CUDAGLBuffer, cwl/include/cwl/buffer.h:88-143): register an OpenGL buffer object, map it and return the
 * device pointer the renderer can write AOVs to; unregister unmaps.  A current OpenGL context is required on the calling thread. */
int fh_gl_register_buffer(fh_ctx* ctx, unsigned int gl_buffer, void** resource, void** device_ptr, uint64_t* bytes);
int fh_gl_unregister_buffer(fh_ctx* ctx, void* resource);

/* -- device memory helpers (stand in for cwl::CUDABuffer, cwl/include/cwl/buffer.h:18-85) */
int fh_malloc(fh_ctx* ctx, uint64_t bytes, void** out);
int fh_free(fh_ctx* ctx, void* ptr);
int fh_memset(fh_ctx* ctx, void* ptr, int value, uint64_t bytes);
int fh_copy_to_device(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes);
int fh_copy_to_host(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes);
/* host-side image decoding for front ends without a decoder of their own (PNG, baseline JPEG, binary PPM/PGM through
 * include/fredholm/image_io.h; stb_image's role in fredholm/src/scene.cpp:7-37).  No context and no GPU needed.  *rgba8 holds
 * width*height*4 bytes, row 0 first (after the optional vertical flip), and is released with fh_image_free.  Returns FH_OK or
 * FH_E_INVALID; the message is available from fh_last_error(NULL). */
int fh_image_load_rgba8(const char* path, int flip_vertically, uint32_t* width, uint32_t* height, uint8_t** rgba8);
void fh_image_free(uint8_t* rgba8);
int fh_copy_on_device(fh_ctx* ctx, void* dst, const void* src, uint64_t bytes); /* asynchronous, ordered on the context stream (cwl::CUDABuffer device-to-device copies) */
void* fh_stream(fh_ctx* ctx); /* hipStream_t of the context */

/* -- batch ray queries over the built BVH (closest hit, or any hit): rays7 = o.xyz d.xyz tmax per ray (host memory); tuv: 3 floats, prim: face id or
 * 0xffffffff (host memory).  The same traversal code the render kernels run; what a caller without a renderer (picking, visibility probes) uses. */
int fh_trace_rays(fh_ctx* ctx, uint32_t n, const float* rays7, int any_hit, float* tuv, uint32_t* prim);
/* what the runtime reports for the streaming traversal kernel the current scene would be traced by (which = 0: closest hit, 1: secondary rays):
 * out[0] = vector registers per lane, out[1] = static LDS bytes per workgroup, out[2] = scratch bytes per lane, out[3] = workgroups per CU the kernel is
 * launched with, out[4] = stack levels it keeps in LDS (the deeper ones spill to global memory), out[5] = stack levels the BVH needs.  Valid after a
 * BVH build; out[3..4] after the first fh_render of the scene (0 before).  Lets a profile be tied to the code object that produced it (bench.py). */
/* which = 2 + c: the shade kernel of shading class c of the uploaded scene (FH_E_INVALID beyond the scene's classes): out[0..2] as above, out[3] = resident workgroups per
 * CU (of 4 waves: = waves per SIMD), out[4] = the lobe mask the kernel is compiled for, out[5] = the lobe mask of the class. */
int fh_kernel_info(fh_ctx* ctx, int which, uint32_t out[6]);
/* measured HBM bandwidth of this GPU (GB/s): a streaming float4 read and a float4 copy (read + written bytes) over `bytes`-sized buffers,
   `iters` launches each.  The "measured HBM roofline" SURVEY.md 8(d) asks for; use buffers well beyond the 256 MiB Infinity Cache. */
int fh_measure_bandwidth(fh_ctx* ctx, uint64_t bytes, uint32_t iters, double* read_gbs, double* copy_gbs);

#ifdef __cplusplus
}
#endif
#endif
