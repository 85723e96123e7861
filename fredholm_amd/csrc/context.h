// context.h -- host-side state behind the opaque fh_ctx of include/fredholm_hip.h.
// Plays the role of the reference's Renderer members (fredholm/include/fredholm/renderer.h:739-828):
// scene buffers, acceleration structure, lights, environment state, sample-count buffer, stream.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/fredholm_hip.h"
#include "fh_device.h"

struct fh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;   // main stream: everything the caller can observe is ordered on it
  hipStream_t aux_stream[2] = {nullptr, nullptr};  // passes j % n_slots != 0 of fh_render run here, overlapping the latency-bound ends of the passes before
  std::string err;
  uint32_t flags = 0;

  // constant tables
  uint32_t* d_sobol = nullptr;
  uint32_t* d_sobol_bytes = nullptr;  // [1024 dims][4 index bytes][256]: byte-indexed generator matrices (capi.hip: fh_ctx_create)
  float* d_lut_refl = nullptr;
  float* d_lut_sheen = nullptr;

  // host copy of the flat scene (transforms can change: Renderer::set_time)
  std::vector<float> h_vertices, h_normals, h_texcoords;
  std::vector<uint32_t> h_indices, h_material_ids, h_instance_ids;
  std::vector<fh_material> h_materials;
  std::vector<float> h_o2w, h_w2o;  // 12 floats per instance
  bool scene_loaded = false, bvh_valid = false;

  // device scene
  float4* d_face_rec = nullptr;
  uint8_t* d_face_cls = nullptr;
  fh::MaterialDev* d_materials = nullptr;
  fh::AreaLightDev* d_lights = nullptr;
  // object-space geometry resident on the device: the face records above are recomputed from it when the instance transforms change
  float* d_obj_vertices = nullptr;
  float* d_obj_normals = nullptr;
  float* d_obj_texcoords = nullptr;
  uint32_t* d_obj_indices = nullptr;
  uint2* d_face_meta = nullptr;  // (material id, instance id) per face
  float4* d_o2w = nullptr;       // 3 rows per instance
  float4* d_w2o = nullptr;
  uint32_t n_xf_alloc = 0;
  // textures (renderer.h:372-386): one device blob of texels + descriptors + the sRGB table
  // per texture: can a filtered fetch of its alpha channel / of its red channel come out below the any-hit threshold 0.5 (upload_textures)?
  std::vector<uint8_t> h_tex_alpha_cuts, h_tex_red_cuts;
  std::vector<fht_texture> h_tex_desc;  // the descriptors as uploaded (device pointers)
  struct HostTexture { uint32_t width = 0, height = 0, srgb = 0; std::vector<uint8_t> rgba8; };
  std::vector<HostTexture> h_tex_host;  // texels of the textures that can cut, kept on the host: per-face opacity classes (capi.hip: footprint_class)
  unsigned long long alpha_cell_counts[3] = {0, 0, 0};  // micromap cells of the faces that keep their test; of them: always pass, never pass (fh_alpha_cell_counts)
  uint32_t alpha_face_counts[4] = {0, 0, 0, 0};  // faces whose textures can cut; of them: always pass, never pass, still tested (fh_alpha_face_counts)
  uint4* d_alpha_rec = nullptr;         // 8 x 16 bytes per face when the scene has cut-outs: what the any-hit test of that face reads (4) and the face's opacity micromap (4) (fh_trace.h: alpha_pass)
  uint8_t* d_texels = nullptr;
  fht_texture* d_textures = nullptr;
  float* d_srgb_lut = nullptr;
  uint32_t n_textures = 0;
  bool has_alpha = false;
  float* d_ibl = nullptr;
  uint32_t ibl_w = 0, ibl_h = 0;
  uint32_t n_faces = 0, n_lights = 0, n_materials = 0;
  uint32_t n_classes = 0;
  uint32_t class_lobes[fh::kMaxClasses] = {};

  // BVH
  float4* d_bvh2_nodes = nullptr;
  float4* d_bvh2_tris = nullptr;
  uint32_t bvh2_n_nodes = 0, bvh2_n_tris = 0;
  uint4* d_bvh8_nodes = nullptr;
  float4* d_bvh8_tris = nullptr;
  uint32_t bvh8_n_nodes = 0, bvh8_n_tris = 0;
  uint32_t* d_bvh8_parent = nullptr;  // per wide node: parent << 3 | child slot in the parent (root: 0xffffffff)
  uint32_t* d_face_node = nullptr;    // per face: the wide node that holds it (bottom-up start of the rays that leave it, fh_trace.h)
  // kept after a full build so that a change of instance transforms refits the wide tree instead of rebuilding it (bvh_build.hip)
  float4* d_bvh8_box = nullptr;             // full-precision (lo, hi) of every wide node
  std::vector<uint32_t> bvh8_level_start;   // node index range of every level (levels are contiguous: the collapse is breadth first)
  bool refit_ok = false;                    // topology unchanged since the last full build, no split references
  double bvh8_area_built = 0.0;             // sum of the node areas right after the full build (quality reference for refits)
  uint32_t n_refits = 0;
  uint32_t bvh8_depth = 0;            // node LEVELS of the wide tree; a traversal stack needs levels - 1 entries (fh_trace.h: stack_entries_for), of which the streaming kernels keep the first in LDS
  uint32_t occupancy_key = 0xffffffffu, occupancy_blocks = 0, occupancy_blocks_secondary = 0;  // resident workgroups per CU of the streaming kernels, as the runtime reports them (render.hip)
  uint32_t info_blocks[2] = {0, 0}, info_entries[2] = {0, 0};  // the same two facts for the uninstrumented closest-hit / secondary kernels as last launched (fh_kernel_info)
  uint32_t lds_configured_bytes = 0;  // dynamic-LDS size the traversal kernels were last configured for (render.hip)
  uint32_t stream_lds_entries = 0, stream_lds_entries_secondary = 0;    // stack levels the closest-hit / secondary streaming kernel keeps in LDS (the rest spills to d_stack_spill)
  uint2* d_stack_spill = nullptr;     // [6 launches in flight][entry beyond the LDS part][thread of the launch]
  size_t stack_spill_capacity = 0;    // in uint2
  uint32_t lds_static_max = 0;        // largest static LDS of a kernel that keeps its traversal stack in dynamic LDS (hipFuncGetAttributes; render.hip: configure_traversal_lds)
  bool use_bvh8 = false;
  int builder_choice = 0;  // 0 = not decided for this scene, 1 = radix tree (LBVH), 2 = PLOC; decided at the first build after an upload
  double bvh_build_ms = 0.0;
  float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};  // padded world bounds of the geometry

  // frame state
  uint32_t width = 0, height = 0;
  uint32_t* d_sample_count = nullptr;   // samples accumulated per pixel (written by k_accumulate)
  uint32_t* d_sample_issued = nullptr;  // samples started per pixel (written after k_generate): the next pass does not wait for the accumulate
  uint32_t shard_rank = 0, shard_world = 1, tile_w = 32, tile_h = 32;
  uint32_t* d_owned = nullptr;  // image indices of owned pixels, in tile order
  uint32_t* d_owned_xy = nullptr;  // the same pixels as x | y << 16
  uint32_t n_owned = 0;
  // pixels that cannot see the scene (render.hip: k_split_pixels / k_sky_pixels): [0] / [1] image indices and x | y << 16 of the pixels the passes render, [2] / [3] of the sky pixels
  uint32_t* d_split[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t split_capacity = 0, n_wave_px = 0, n_sky_px = 0;
  uint32_t* d_split_counters = nullptr;  // wave pixels, sky pixels, bounds-test violations seen by k_sky_pixels, pad
  bool split_valid = false;
  float split_key[32] = {};              // camera, scene bounds, resolution and ownership the lists were made for
  hipStream_t sky_stream = nullptr;
  hipEvent_t ev_sky = nullptr;
  struct ShardList { uint32_t rank, world, width, height, tile_w, tile_h; uint32_t* d_owned; uint32_t n_owned; };
  // fh_unpack_shards: the ownership lists of all ranks of a split one after the other (a permutation of the frame) and where each rank's begins
  struct FrameMap { uint32_t world = 0, width = 0, height = 0, tile_w = 0, tile_h = 0; uint32_t* d_all = nullptr; std::vector<uint32_t> start; };
  FrameMap frame_map;
  std::vector<ShardList> shard_lists;  // ownership lists of other ranks' shards, built on first use by fh_unpack_shard and kept (freed with the context)

  // environment (renderer.h:819-827)
  bool has_dir = false;
  float dir_le[3] = {0, 0, 0}, dir_dir[3] = {0, 1, 0}, dir_angle = 0;
  float sky_intensity = 1.0f;
  float sun_dir[3] = {0.0f, 1.0f, 0.0f};
  bool has_hosek = false;
  fh::HosekSky hosek{};
  fh::HosekSky* d_hosek = nullptr;  // device copy (FrameDev::hosek)

  // path pools: one per pass in flight (pass j uses slot j % n_slots and the stream of that slot); allocated on first use
  fh::PoolDev pool[3] = {};
  struct PoolShape { bool dir = false, lights = false; uint32_t classes = 0; };  // what the records of a pool have room for (render.hip: pool_ensure)
  PoolShape pool_shape[3];
  std::vector<void*> pool_allocs[3];
  unsigned long long pool_alloc_bytes[3] = {0, 0, 0};  // device memory the allocations of each pool hold (fh_path_pool_allocated)
  // 32 Mi path slots per pool: 16 samples per pixel per pass at 1080p.  Three pools (one per pass in flight) of 284-436 bytes per path are 27-42 GB when a call brings enough
  // samples to fill them; unless the caller chose the size (fh_set_path_pool), fh_render keeps all pools together within half of the device memory that is free when the first one is made
  uint32_t pool_target_default = 1u << 25, pool_target = 1u << 25;  // (pool_target: the default as capped by the free device memory whenever a pool is (re)allocated, or the caller's size)
  bool pool_target_by_caller = false;
  uint32_t tail_depth = 0;          // bounces run as wavefront kernels before k_tail finishes the survivors; 0 = adaptive
  uint32_t auto_wave_depth = 2;     // adaptive choice of the pass submitted last (kept for the debug print)
  // the adaptive choice is made per pass from the SHARE of a pass's paths still alive at each depth, taken from the newest snapshot of an earlier pass's counters: a share is a
  // property of the scene and the camera, not of the pass, so a call of another size (one final frame of 4096 spp after a session of 1-spp calls, or the other way round)
  // picks its depth right from its first pass.  (Up to round 5 the DEPTH picked for the earlier pass was carried over: the first 16-spp passes after small calls handed
  // millions of paths to the fused tail, 300 ms each on configs[3].)  survival_n = 0: nothing known yet (a new tree: bvh_build.hip); entries deeper than the newest
  // snapshot's wavefront bounces are an earlier pass's.
  double survival[66] = {};
  uint32_t survival_n = 0;
  uint32_t* h_counters[3] = {nullptr, nullptr, nullptr};  // pinned snapshots of the per-bounce counters of a finished pass
  hipEvent_t ev_counters[3] = {nullptr, nullptr, nullptr};
  bool counters_in_flight[3] = {false, false, false};
  uint32_t counters_wave_depth[3] = {0, 0, 0};  // wave depth used by the pass the snapshot comes from
  uint32_t counters_paths[3] = {0, 0, 0};       // ... and the paths it started
  int counters_bu[3] = {-1, -1, -1};            // ... and, while the scene is being probed, where its first-hit rays started (0: the root, 1: the node of their face; -1: not a probing pass)
  int bu_choice = 0;                            // 0: probing, 1: rays start at the root, 2: first-hit rays start at the node of the face they leave (render.hip; reset by every full BVH build)
  uint32_t bu_toggle = 0;
  double bu_cost[2] = {0.0, 0.0}, bu_items[2] = {0.0, 0.0};
  unsigned long long pass_seq = 0;           // passes submitted so far
  hipEvent_t ev_gen[3] = {nullptr, nullptr, nullptr}, ev_acc[3] = {nullptr, nullptr, nullptr}, ev_enter = nullptr;
  bool gen_valid[3] = {false, false, false}, acc_valid[3] = {false, false, false};
  int n_slots = 3;  // passes in flight (FH_PIPELINE=0: 1, every pass on the main stream; =2: two).  Three against two: +2.3 % on configs[2], +0.3-0.9 % on the others, one more path pool
  int last_slot_used = 0;  // slot of the pass submitted last (what the next pass orders itself after)
  std::vector<hipEvent_t> ev_bounce;  // single-pass calls: (shade done, secondary done) per bounce, for the secondary launch on a second stream (render.hip)
  // FH_FLAG_REFERENCE_FIRSTHIT (render.hip: k_firsthit_scan): per-pixel "a sample of this launch has hit something" + the AOVs of that hit
  uint32_t* d_quirk_seen = nullptr;
  float4* d_quirk_aov = nullptr;
  size_t quirk_pixels = 0;

  // device facts and developer switches, read ONCE at fh_ctx_create (fh_render does no getenv / hipGetDeviceProperties)
  struct Tunables {
    uint32_t n_cus = 256;
    uint32_t lds_per_cu = 160u * 1024u, lds_per_block = 160u * 1024u;  // LDS of a CU / the most one workgroup may take, from the device attributes (gfx950: 160 KB both)
    uint32_t coop_flush = 48;       // FH_COOP_T: queued candidate triangles that trigger a cooperative test round (r5-12: 32 -> 48, configs[3] +1.6 %, configs[2] +0.5 %)
    bool coop_flush_fixed = false;  // FH_COOP_T was given: no per-call choice (render_submit)
    bool coop = true;               // FH_COOP=0: per-lane triangle loop
    bool stream = true;             // FH_STREAM=0: one fixed batch per wave; FH_STREAM=1: streaming whatever the size of the tree
    bool stream_forced = false;
    uint32_t stream_refill = 24;    // FH_STREAM_REFILL: idle lanes that trigger a refill
    uint32_t stream_min_rays = 64;  // FH_STREAM_MIN_RAYS: queue entries per wave below which workgroups of a streaming launch stay out (render.hip: stream_block_idle); 0 = all take part
    bool overlap_secondary = true;  // FH_OVERLAP=0: single-pass calls keep every launch on one stream
    bool sky_split = true;          // FH_SKY_SPLIT=0: every pixel goes through the passes (no k_sky_pixels)
    uint32_t sky_split_min_log2 = 27; // FH_SKY_SPLIT_MIN_LOG2: a call splits its pixels from 2^n camera paths on (27: 64 spp of a 1080p frame; tests lower it)
    uint32_t sky_blocks_per_cu = 0; // FH_SKY_BLOCKS: workgroups per CU k_sky_pixels is launched with (grid-stride over the sky pixels); 0 = one thread per pixel
    bool poison_pools = false;      // FH_POISON=1: new path pools are filled with 0xa5 before their first use (tests: nothing may read what nobody wrote)
    bool merge_trace = true;        // FH_MERGE=0: single-pass calls trace secondary rays and the next bounce's closest-hit rays in two launches (two streams) instead of one
    uint32_t shade_wgs = 0;         // FH_SHADE_WGS=2|3: workgroups per CU the shade kernels are compiled for (0: three; until r5-12 for textured scenes that stream only)
    uint32_t stack_lds_entries = 0; // FH_STACK_LDS=n: stack levels the streaming kernels keep in LDS (0: as many as cost no workgroup; 99: all)
    bool sort_small = false;        // FH_SORT_SMALL=1: cell-order the bounce queues of trees the fixed-batch kernels trace as well
    uint32_t stream_chunk = 64;     // FH_STREAM_CHUNK: queue entries a wave takes per global atomic (setting it also switches the adaptive maximum off)
    bool stream_chunk_fixed = false;
    uint32_t stream_chunk_closest = 128; // FH_STREAM_CHUNK_CLOSEST: the same for the closest-hit launch (0 = stream_chunk).  Its queue is in cell order, so a longer run of it is a more coherent
                                         // wave: 64 / 96 / 128 / 192 / 256 / 512 entries measure 78.5 / 74.0 / 74.2 / 75.7 / 76.7 / 84.1 ms per configs[2] frame; configs[4] 2795 -> 2559 ms.  (The secondary launch,
                                         // whose items are whole paths with two to four rays each, loses with more than 64: 113.9 -> 115.7 ms at 128)
    uint32_t tail_depth = 0;        // FH_TAIL_DEPTH: fixed number of wavefront bounces before k_tail
    uint32_t tail_paths = 0;        // FH_TAIL_PATHS: survivors at which the adaptive mode switches to k_tail; 0 = 65536, 262144 for passes of at most 4 Mi paths
    uint32_t bottom_up = 2;         // FH_BOTTOM_UP=0: every ray starts its traversal at the root; =1: first-hit rays of scenes without cut-outs start at the wide node that holds the face
                                    // they leave and climb; default (2): the first passes after a build try both and the counted test rounds per shaded path decide (render.hip)
    bool sort_queues = true;        // FH_SORT=0: trace the bounce queues in emission order
    uint32_t sort_onepass = 2;      // FH_SORT_ONEPASS: cell sorts in calls of ONE pass -- 0 none, 1 all (as in multi-pass calls), 2 only the queue the fused tail takes over
    bool debug_tail = false;        // FH_DEBUG_TAIL
    bool force_alpha = false;       // FH_FORCE_ALPHA=1 (timing experiments): the kernels with the any-hit path compiled in, whatever the scene
  } tun;

  // bloom weights of the last sigma used (post.hip): no allocation, upload or host synchronisation per frame
  float* d_bloom_weights = nullptr;
  float bloom_sigma_cached = -1.0f, bloom_wsum = 0.0f;

  // denoiser slot (post.hip): ping-pong buffers of the a-trous filter
  float4* d_denoise_tmp[2] = {nullptr, nullptr};
  size_t denoise_pixels = 0;

  // stats
  fh_stats stats{};
  unsigned long long* d_trace_counters = nullptr;  // nodes, tris, rays of the closest-hit kernel, then of the secondary kernel
  struct TimedSpan { hipEvent_t a, b; int kind; };
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t ev_render_begin = nullptr, ev_render_end = nullptr;
  bool render_pending = false;
};

namespace fh {
int fail(fh_ctx* ctx, int code, const std::string& msg);
#define FH_HIP(call)                                                                                                   \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess) return fh::fail(ctx, FH_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
  } while (0)

SceneDev scene_dev(const fh_ctx* ctx);
int bvh_build_device(fh_ctx* ctx);                 // bvh_build.hip
int render_submit(fh_ctx* ctx, const fh_camera* cam, const float* bg, const fh_render_layers* layers, uint32_t n_samples, uint32_t max_depth, uint32_t seed);  // render.hip
int pool_ensure(fh_ctx* ctx, int slot, uint32_t capacity);   // render.hip
uint64_t pool_bytes_per_path(const fh_ctx* ctx);            // capi.hip
void pool_release(fh_ctx* ctx);
int kernel_info(fh_ctx* ctx, int which, uint32_t out[6]);   // render.hip
int post_process_submit(fh_ctx* ctx, const float* in, float* hi, float* tmp, int w, int h, const fh_post_params* pp, float* out);  // post.hip
int denoise_submit(fh_ctx* ctx, int w, int h, const float* beauty, const float* normal, const float* albedo, float* out, int upscale);  // post.hip
}  // namespace fh
