// fh_device.h -- device-visible data layout of the wavefront path tracer (all plain structs passed
// to kernels by value as kernel arguments).
//
// HBM layout:
//   face records   7 x float4 per face, face-indexed, world space (positions, inverse-transpose
//                  normals, uvs, material id) -- what fill_surface_info (pt.cu:141-179) gathers
//                  through three index/vertex/normal/texcoord arrays becomes one 112-byte record.
//   BVH            wide nodes + leaf triangles in traversal order (fh_trace.h).
//   path pool      structure-of-arrays indexed by path slot p (one slot per camera path of the pass).
//   queues         uint32 path-slot lists with device-resident counters; no host readback inside a frame.
#pragma once
#include "../../include/fh_texture_unit.h"
#include "fh_sky.h"
#include "fh_vec.h"

namespace fh {

// float4s per face record: positions, normals, texture coordinates in the .w lanes, material, start node.  (Padding a record to one 128-byte line -- 8 -- and handing k_shade
// the face id with its queue entry, so that it need not wait for the hit record, were both measured in round 6 and buy nothing: tools/patches/r6_face_rec_128_and_cls_prim.patch)
constexpr uint32_t kFaceRec = 7;
constexpr uint32_t kMaxShardsPerLaunch = 16;  // fh_unpack_shards: ranks whose packed shards one launch un-permutes (a node has 8)
constexpr uint32_t kMaxClasses = 8;       // shading classes (distinct lobe masks) per scene
constexpr uint32_t kNumQueues = kMaxClasses;

// secondary-ray slots per shaded path, in the reference's evaluation order (pt.cu:772-925)
enum : uint32_t { SEC_DIR = 0, SEC_SKY = 1, SEC_AREA = 2, SEC_LIGHT = 3, SEC_COUNT = 4 };

struct BsdfTables { const float* reflection; const float* sheen; };  // 16x16x2 and 16x16 floats (lut.cu:5-93, :917-955)

struct MaterialDev {  // 180-byte reference record + derived words
  float w[45];
  uint32_t lobes;     // lobe mask of this material (fh_bsdf.h)
  uint32_t emissive;  // has_emission (pt.cu:125-129)
  uint32_t alpha;     // base-colour or alpha texture present: candidate hits go through the any-hit test
  uint32_t cls;       // shading class index
};

struct AreaLightDev { uint32_t face; uint32_t material; };

struct Bvh2Dev {
  const float4* nodes;  // 4 x float4 per node (two child boxes + two child refs)
  const float4* tris;   // 3 x float4 per leaf triangle, traversal order; tris[3i].w = face id bits
  uint32_t n_nodes;
  uint32_t n_tris;
};

// wide BVH (8 children, quantised boxes), see fh_trace.h
struct Bvh8Dev {
  const uint4* nodes;   // 4 x uint4 (64 bytes) per node
  const float4* tris;   // 3 x float4 per triangle slot, eight slots per node (slot 8 * node + child slot)
  uint32_t n_nodes;
  uint32_t n_tris;
  uint32_t depth;       // entries of the LDS traversal stack: levels of the tree - 1 (fh_trace.h: stack_entries_for)
  const uint32_t* parent;  // per node: parent << 3 | child slot in the parent, root 0xffffffff; null: rays start at the root (fh_trace.h: bottom-up start)
};

struct SceneDev {
  const float4* face_rec;       // kFaceRec per face
  const uint8_t* face_cls;      // shading class of the face's material | 0x80 if emissive
  const MaterialDev* materials;
  const AreaLightDev* lights;
  uint32_t n_faces, n_lights, n_materials;
  const fht_texture* textures;  // software texture unit (include/fh_texture_unit.h)
  const float* srgb_lut;        // 256-entry sRGB -> linear table
  uint32_t n_textures;
  const uint4* alpha_rec;       // per face (8 vectors), scenes with cut-outs only: (uv0, uv1), (uv2, flags), (alpha-carrying base texture), (alpha texture), 4 x opacity micromap
  uint32_t has_alpha;           // some faces carry an alpha cut-out (pt.cu:545-678): traversal runs the any-hit test for them
  Bvh2Dev bvh2;
  Bvh8Dev bvh8;
  uint32_t use_bvh8;
  const uint32_t* face_node;    // per face: the wide node that holds it; null when rays start at the root
};

struct FrameDev {
  uint32_t width, height;
  uint32_t seed_hash;  // xxhash32(seed), used as Sobol seed and CMJ scramble (pt.cu:388,393)
  uint32_t max_depth;
  uint32_t has_dir, has_hosek, has_ibl;
  fht_texture ibl;  // float4 lat-long environment (renderer.h:574-581)
  uint32_t n1, n2;     // Sobol dimensions / CMJ slots consumed per shaded bounce
  // camera (camera.cu:24-53)
  m34 cam_xf;
  float cam_inv_tan, cam_F, cam_focus;
  float cam_a_plus_b, cam_lens_radius;  // a + b with a = 1 / (1 + f - 1 / b), and 2 f / F (camera.cu:33-36): per-frame constants, formed on the host
  // environment
  f3 bg;
  float sky_intensity;
  f3 sun_dir;
  const HosekSky* hosek;  // 30 floats in device memory (fh_ctx::d_hosek); kernels that evaluate the sky per path stage them in LDS (render.hip: stage_sky) -- as a kernel argument they sat in 30 SGPRs next to everything else and the compiler spilled scalars into vector lanes (k_generate: 162 v_readlane / v_writelane)
  f3 dir_le, dir_dir;
  float dir_disk_radius;  // 1e9 * tan(rad(angle/2)), pt.cu:333-335
  // padded scene bounds: camera rays that miss them skip the traversal queue
  f3 scene_lo, scene_hi;
  f3 cell_scale;  // 2^kCellBits / (scene_hi - scene_lo)
  // tables
  const uint32_t* sobol;        // 1024 x 52
  const uint32_t* sobol_bytes;  // 1024 x 4 x 256, byte-indexed form
  BsdfTables lut;
};

// A path's state is kept in records, not in one array per field: after the first bounce the queues hold path slots in spatial order, so every access to
// slot p is a random one and a kernel that touches n fields of a path touched n cache lines.  With records the closest-hit kernel reads a ray and writes its
// hit in ONE 64-byte line, the shade kernel finds ray, throughput and hit in that same line, and the up to four secondary rays of a path are contiguous.
// Rec<T, STRIDE>[p] addresses field T of record p (STRIDE in units of T), so kernels keep writing pool.ray_o[p].
template <typename T, uint32_t STRIDE>
struct Rec {
  T* base;
  FH_HD T& operator[](size_t p) const { return base[p * STRIDE]; }
};

struct PoolDev {
  uint32_t capacity;
  // 64-byte path record
  Rec<float4, 4> ray_o;  // origin.xyz, tmax
  Rec<float4, 4> ray_d;  // direction.xyz, -
  Rec<float4, 4> thr;    // throughput.xyz, -
  Rec<float4, 4> hit;    // t, u, v, face id bits (0xffffffff = miss)
  float4* rad;           // radiance.xyz, -   (its own array: k_accumulate streams it)
  // 8-byte identity record
  Rec<uint32_t, 2> pixel;
  Rec<uint32_t, 2> nspp;
  uint32_t* flags;  // bit0: first-hit AOVs valid, bit1: path ended before its first ray (Russian roulette draw of 1.0)
  // first-hit AOV staging (pt.cu:745-751), one 64-byte record
  Rec<float4, 4> aov_position;
  Rec<float4, 4> aov_normal;
  Rec<float4, 4> aov_albedo;
  Rec<float4, 4> aov_texdepth;  // texcoord.xy, depth, -
  // secondary rays: SEC_COUNT x 3 float4 per path -- origin.xyz + tmax, direction.xyz + active (1.0f) / inactive (0.0f), contribution rgb if unoccluded
  float4* sec;
  // only the ray kinds the scene can emit have a place in the record (no directional light, no emitters: sky ray + light ray = 96 bytes instead of 192)
  uint32_t sec_count;       // places per path
  uint32_t sec_index;       // place of slot s: byte s of this word
  FH_HD size_t sec_at(uint32_t slot, uint32_t p) const { return ((size_t)p * sec_count + ((sec_index >> (8u * slot)) & 0xffu)) * 3u; }
  // BSDF-sampled light ray when the scene has emitters (needs the hit to finish the MIS weight), one 32-byte record
  Rec<float4, 2> lp_a;   // throughput.xyz, |cos|
  Rec<float4, 2> lp_b;   // f.xyz, pdf
  // queues
  uint32_t* q_rad[2];            // radiance-ray queue, ping-pong per bounce
  uint32_t* q_cls;               // (shading classes of the scene) x capacity: hits routed by shading class
  uint32_t* q_sec;               // shaded paths with secondary rays
  uint32_t* q_prim;              // face id of the closest hit of radiance-queue ENTRY i of the bounce being traced (0xffffffff: miss), written next to pool.hit by whoever traces
                                 // entry i: k_route then reads queue and face ids as two streams instead of one 64-byte path record per ray for four of its bytes
  uint32_t* counters;            // kCounterStride words per bounce, zeroed once per pass
  // spatial ordering of the bounce queues (render.hip: sort_queue_by_cell): the shade kernel stores the cell of the hit point next
  // to every queue entry it appends; a counting sort brings entries of one cell together before the rays are traced
  uint16_t* key_sec;             // cell of q_sec[i]
  uint16_t* key_rad;             // cell of the entries appended to the next bounce's radiance queue
  uint32_t* q_tmp;               // third radiance queue: sorted output rotates with the ping-pong pair
  uint32_t* q_sec_sorted;
  uint32_t* bins;                // 2 x kCells words: histogram (left zero by the scan), cursors
};
constexpr uint32_t kCellBits = 4;                      // per axis
constexpr uint32_t kCells = 1u << (3 * kCellBits);     // 4096 cells over the scene bounds, Morton-numbered
// per-bounce counter block: everything a bounce produces or consumes has its own word, so one
// memset per pass replaces per-bounce resets
enum : uint32_t {
  CNT_RAD = 0,        // entries of the radiance queue consumed by this bounce
  CNT_SEC = 1,        // shaded paths with secondary rays
  CNT_CUR_CLOSEST = 2,  // work cursor of the persistent closest-hit kernel
  CNT_CUR_SEC = 3,      // work cursor of the persistent secondary kernel
  CNT_CLS = 4,        // kMaxClasses class-queue counts
  CNT_COST_NODE = 12, // wave-level node-test rounds of this bounce's secondary (or merged) streaming launch, summed over its waves ...
  CNT_COST_TRI = 13,  // ... and its wave-level triangle-test rounds: what the library weighs rays that start at the root against rays that start at their face with (render.hip)
  kCounterStride = 16,
};
static_assert(CNT_CLS + kMaxClasses <= kCounterStride, "counter block too small");

struct LayersDev {
  float4* beauty; float4* position; float* depth; float4* normal; float4* texcoord; float4* albedo;
  uint32_t* sample_count;
};

// traversal statistics (only written by the instrumented kernel variants)
struct TraceCounters { unsigned long long* nodes; unsigned long long* tris; unsigned long long* rays; unsigned long long* wave_nodes; unsigned long long* wave_tris; unsigned long long* hist;  // hist: 8 buckets of node steps per ray (<=8, <=16, ... <=512, more)
                       unsigned long long* clk; };  // FH_FLAG_TIME_KERNELS, streaming kernels: summed shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of their waves, else null

// clock the chip holds while a kernel runs (MI355X_MICROARCH.md, DVFS give-back item 6): every wave stamps both counters at its start and its end
struct ClockStamp {
  unsigned long long t0, r0;
  FH_D ClockStamp() : t0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
  FH_D void commit(unsigned long long* clk) const
  {
    if (!clk) return;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (__lane_id() == 0u) { atomicAdd(clk, t1 - t0); atomicAdd(clk + 1, r1 - r0); }
  }
};

// ---- wave-aggregated queue append: one atomic per wave (ballot + popcount prefix)
// same append, storing a 16-bit key at the same position of a parallel array
FH_D void queue_push_keyed(uint32_t* counter, uint32_t* queue, uint16_t* keys, bool active, uint32_t value, uint32_t key)
{
  const unsigned long long mask = __ballot(active);
  if (mask == 0ull) return;
  const uint32_t lane = __lane_id();
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (active) {
    const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    queue[pos] = value;
    keys[pos] = (uint16_t)key;
  }
}

// ---- block-aggregated queue append: ONE returning atomic per workgroup and call instead of one per wave.  A counter is one address, and the chip serves a bounded
// number of returning atomics per address and second: with one per wave, kernels whose waves are short (k_generate: all camera rays of an interior enter the scene) or
// that append to two queues per wave (k_shade) waited for their queue positions.  N appends at once (one ballot each), all 256 threads of the workgroup call together;
// `scratch` is N x (4 wave counts + 1 base) words of LDS that the NEXT call may reuse only after a workgroup barrier (the callers' loops have one, or alternate two areas).
template <int N>
FH_D void block_queue_reserve(uint32_t* const (&counter)[N], const bool (&active)[N], uint32_t (&pos)[N], uint32_t* scratch)
{
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long m[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    m[k] = __ballot(active[k]);
    if (lane == 0u) scratch[5 * k + wave] = (uint32_t)__popcll(m[k]);
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)N) {
    uint32_t* c = scratch + 5 * threadIdx.x;
    uint32_t total = 0;
    for (uint32_t w = 0; w < 4u; ++w) { const uint32_t v = c[w]; c[w] = total; total += v; }
    c[4] = total ? atomicAdd(counter[threadIdx.x], total) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) pos[k] = scratch[5 * k + 4] + scratch[5 * k + wave] + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
}

FH_D void queue_push(uint32_t* counter, uint32_t* queue, bool active, uint32_t value)
{
  const unsigned long long mask = __ballot(active);
  if (mask == 0ull) return;
  const uint32_t lane = __lane_id();
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (active) queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

}  // namespace fh
