// render.hip -- the wavefront path tracer: kernels + per-frame submission.
//
// Replaces Renderer::render -> optixLaunch of the megakernel in fredholm/modules/pt.cu
// (__raygen__rg :418-502, __closesthit__radiance :680-944, __closesthit__light :952-999,
// __miss__* :504-543).  One reference launch thread = one pixel looping over samples and bounces;
// here a pass starts `B` samples for every owned pixel as independent path slots and advances all
// of them one bounce at a time:
//
//   k_generate               CMJ slots 0/1 -> thin-lens camera ray, throughput 1; rays that miss the scene bounds are
//                            finished here (sky / background) and never enter a queue            (pt.cu:433-454, :504-523)
//   k_trace_closest_stream   closest hit for the radiance-ray queue: waves draw chunks of the queue, refill idle lanes in
//                            flight and test candidate triangles cooperatively (fh_trace.h: traverse_stream)
//   k_route                  sorts the hits into one queue per shading class (BSDF-sorted shading) with block-aggregated
//                            appends: ballot + popcount per wave, LDS prefix per block, one atomic per block and class
//   k_shade<LOBES>           surface + BSDF + NEE samples + light ray + next direction + Russian roulette for the next bounce;
//                            emits secondary rays with their pre-weighted contributions and the cell key of the hit point
//   k_miss_primary           sky / background for paths that leave the scene at depth 0               (pt.cu:504-523)
//   k_cell_hist/scan/scatter counting sort of the secondary and next-bounce queues by the cell of the ray origin
//   k_trace_secondary_stream any-hit shadow rays (and the BSDF-sampled light ray) of every shaded path, in the reference's
//                            order; adds the contributions that turn out unoccluded
//   k_tail                   all remaining bounces of the few paths still alive after the wavefront bounces, one kernel
//   k_accumulate             NaN guard + running mean of the 6 AOVs, sample_count += B                (pt.cu:474-501)
// The *_static and *_coop kernels are the fixed-batch forms (binary-BVH fallback of tiny scenes, FH_STREAM=0 / FH_COOP=0).
//
// Sampler slots are addressed absolutely (fh_sampler.h): at bounce b the Sobol' dimension base is
// 1 + b*n1 and the CMJ slot base is 2 + b*n2 with n1 = 3 + [lights], n2 = 3 + [directional] + [lights],
// which reproduces the draw order of SURVEY.md appendix A without per-path sampler state.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "context.h"
#include "fh_bsdf.h"
#include "fh_trace.h"

namespace fh {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kLutReflFloats = 16 * 16 * 2, kLutSheenFloats = 16 * 16;  // lut.cu:5-93, :917-955
constexpr uint32_t kMatLds = 32;  // material records staged in LDS by the shade kernels when the scene has at most this many
#ifndef FH_SECONDARY_BLOCKS_HEAVY
#define FH_SECONDARY_BLOCKS_HEAVY 5  // resident workgroups per CU the secondary streaming kernel is compiled for when it carries the emitter / any-hit code
#endif
#ifndef FH_STREAM_BLOCKS
#define FH_STREAM_BLOCKS 7  // resident workgroups per CU (= waves per SIMD) the streaming traversal kernels are compiled and launched for: 512 / that registers per lane (72 at seven), 160 KB / that of LDS
                            // (seven stack levels next to the cooperative-test records; deeper levels spill).  Six / seven / eight measured on the same box (profiles/README.md r4-1): configs[3] 542 / 598 / 539
                            // Msamples/s, configs[2] 6552 / 6702 / 6017 -- at eight the secondary kernel spills registers and keeps four stack levels in LDS
#endif
#ifndef FH_STREAM_BLOCKS_ALPHA
#define FH_STREAM_BLOCKS_ALPHA FH_STREAM_BLOCKS  // the same for the secondary-ray kernel with the any-hit test compiled in
#endif
#ifndef FH_STREAM_BLOCKS_CLOSEST
#define FH_STREAM_BLOCKS_CLOSEST FH_STREAM_BLOCKS  // the same for the closest-hit kernel, which needs fewer registers than the secondary-ray kernel
#endif
#ifndef FH_SHADE_BLOCKS
#define FH_SHADE_BLOCKS 2  // resident workgroups per CU the specialised shade kernels are compiled for (register budget = 512 / that per lane): they take 159-182 registers
#endif                     // since their products go to memory as they are made (PoolSink); the generic seven-lobe kernel keeps one wave per SIMD (350 registers).
// A second set is compiled for THREE workgroups per CU (168 registers, 6-7 of them spilled: 28-44 B of scratch), and it is the set every scene gets (render_submit;
// FH_SHADE_WGS=2|3 forces either).  Round 3 gave it to textured scenes only -- their hits wait for texels and a third wave per SIMD covers that (configs[3]: shade 3.87 -> 3.53 s
// alone; with two, today, 676 -> 644 Msamples/s) -- because the untextured ones lost with it then (configs[1] 2210 -> 2025).  On round 5's build they gain as well
// (profiles/r05_tunables.log: configs[1] 2751 -> 2823, configs[2] 7841 -> 7972, configs[4] with the flush threshold below 5559 -> 5772).

// generator matrices of the N Sobol' dimensions a kernel draws from, staged in LDS in their byte-indexed form (4 KB per dimension, FrameDev::sobol_bytes):
// the XOR over the 32 index bits is four LDS reads instead of 32 bit tests (~85 instructions less per draw; the shade kernels draw three or four per hit)
template <int N>
struct SobolRows {
  uint32_t m[N][1024];
};

template <int N>
FH_D void load_sobol_rows(SobolRows<N>& rows, const uint32_t* tables, const uint32_t* dims)
{
  static_assert(kBlock == 256, "one uint4 per thread and dimension");
#pragma unroll
  for (int r = 0; r < N; ++r)
    reinterpret_cast<uint4*>(rows.m[r])[threadIdx.x] = reinterpret_cast<const uint4*>(tables + (size_t)(dims[r] & 1023u) * 1024u)[threadIdx.x];
  __syncthreads();
}

// the sky coefficients of the frame, copied to the workgroup's LDS; the caller's next barrier publishes them
FH_D void stage_sky(FrameDev& fr, HosekSky& lds_sky)
{
  if (!fr.has_hosek) return;
  if (threadIdx.x < sizeof(HosekSky) / 4u) reinterpret_cast<float*>(&lds_sky)[threadIdx.x] = reinterpret_cast<const float*>(fr.hosek)[threadIdx.x];
  fr.hosek = &lds_sky;
}

// environment seen along d: IBL, else Hosek sky, else the constant background (pt.cu:511-517, :536-542)
FH_D f3 env_radiance(const FrameDev& fr, f3 d)
{
  if (fr.has_ibl) {  // fetch_ibl (pt.cu:344-350) with cartesian_to_spherical (math.cu:111-118)
    const float theta = fhe_acos(clampf(d.y, -1.0f, 1.0f));
    float phi = fhe_atan2(d.z, d.x);
    if (phi < 0) phi += 2.0f * kPi;
    float o[4];
    fht_tex2d(&fr.ibl, nullptr, phi / (2.0f * kPi), theta / kPi, o);
    return fr.sky_intensity * mk3(o[0], o[1], o[2]);
  }
  return fr.has_hosek ? hosek_radiance(*fr.hosek, fr.sun_dir, fr.sky_intensity, d) : fr.bg;
}

// sample n_spp of pixel (px, py): CMJ slots 0 / 1 -> thin-lens camera ray (pt.cu:433-454, camera.cu:24-53) and the Russian roulette of bounce 0, which has probability 1 but
// still consumes (and can fail on) a draw (pt.cu:457-461).  Returns whether the path is alive.
FH_D bool camera_ray(const FrameDev& fr, const uint32_t* sobol_dim1, uint32_t image_idx, uint32_t px, uint32_t py, uint32_t n_spp, f2 u, f2 u_lens, f3& org, f3& dir);
FH_D bool camera_sample(const FrameDev& fr, const uint32_t* sobol_dim1, uint32_t image_idx, uint32_t px, uint32_t py, uint32_t n_spp, f3& org, f3& dir)
{
  return camera_ray(fr, sobol_dim1, image_idx, px, py, n_spp, cmj_draw(n_spp, image_idx, 0u, fr.seed_hash), cmj_draw(n_spp, image_idx, 1u, fr.seed_hash), org, dir);
}
// (u, u_lens: CMJ slots 0 and 1 of the sample)
FH_D bool camera_ray(const FrameDev& fr, const uint32_t* sobol_dim1, uint32_t image_idx, uint32_t px, uint32_t py, uint32_t n_spp, f2 u, f2 u_lens, f3& org, f3& dir)
{
  float uvx = (2.0f * (px + u.x) - fr.width) / fr.height;
  const float uvy = (2.0f * (py + u.y) - fr.height) / fr.height;
  uvx = -uvx;
  u = u_lens;
  // thin lens (camera.cu:24-53); a + b and the lens radius are the same for every ray: computed once on the host, in fp32 with the reference's operations
  const float f = fr.cam_inv_tan;
  const f3 p_sensor = mk3(uvx, uvy, 0.0f);
  const f3 p_lens_center = mk3(0.0f, 0.0f, f);
  const f2 pd = fr.cam_lens_radius * concentric_disk(u);
  const f3 p_lens = p_lens_center + mk3(pd.x, pd.y, 0.0f);
  const f3 s2c = normalize(p_lens_center - p_sensor);
  const f3 p_object = p_sensor + (fr.cam_a_plus_b / s2c.z) * s2c;
  org = xform_point(fr.cam_xf, p_lens);
  f3 d = normalize(p_object - p_lens);
  d.z *= -1.0f;
  dir = xform_dir(fr.cam_xf, d);
  const uint32_t sidx = image_idx + n_spp * fr.width * fr.height;
  const float rr = sobol_draw_bytes(sobol_dim1, sidx, 1u, fr.seed_hash);
  return fr.max_depth > 0 && !(rr >= 1.0f);
}

// ------------------------------------------------------------------------------------------------
// grid: x over the owned pixels (grid-stride), y = sample of the pass -- slot p = sample * n_owned + pixel without a division per path, and the pixel's
// coordinates come packed from the ownership list instead of from image_idx / width and % width (four integer divisions by run-time values were ~60 of the
// kernel's ~1070 non-FMA instructions per path)
constexpr int kGenChunks = 4;  // (eight measure the same: profiles/README.md r4-20)
__global__ void __launch_bounds__(kBlock) k_generate(FrameDev fr, PoolDev pool, const uint32_t* issued, const uint32_t* owned, const uint32_t* owned_xy, uint32_t n_owned)
{
  __shared__ SobolRows<1> rows;
  __shared__ HosekSky s_sky;
  stage_sky(fr, s_sky);
  const uint32_t dims[1] = {1u};
  load_sobol_rows<1>(rows, fr.sobol_bytes, dims);
  // A workgroup takes kGenChunks x 256 consecutive pixels per round and appends the paths that enter the scene with ONE returning atomic (block_queue_reserve): with one per
  // wave, an interior -- every camera ray enters -- spent most of the kernel waiting for queue positions (configs[3]: 188 -> see profiles/README.md r4)
  const uint32_t stride = gridDim.x * blockDim.x * kGenChunks;
  const uint32_t k = blockIdx.y;
  __shared__ uint32_t s_reserve[2][5 * kGenChunks];
  uint32_t iter = 0;
  for (uint32_t base = blockIdx.x * blockDim.x * kGenChunks; base < n_owned; base += stride) {
   uint32_t packed = 0u;  // per chunk one byte: bit 7 = this lane's path enters the scene, bits 0-5 = its rank among the entering lanes of its wave
   uint32_t* const scratch = s_reserve[iter & 1u];
#pragma unroll 1
   for (int c = 0; c < kGenChunks; ++c) {  // (a loop, not four copies: the body is 18 KB of code)
    const uint32_t i = base + (uint32_t)c * blockDim.x + threadIdx.x;
    const bool valid = i < n_owned;
    const uint32_t p = k * n_owned + i;  // slot p = sample-major: lanes of a wave hold neighbouring pixels of one sample index
    bool enter = false;
    if (valid) {
      const uint32_t image_idx = owned[i];
      const uint32_t n_spp = issued[image_idx] + k;  // sample index = samples started on this pixel so far (pt.cu:423: params.sample_count)
      const uint32_t xy = owned_xy[i];
      f3 org, dir;
      const bool alive = camera_sample(fr, rows.m[0], image_idx, xy & 0xffffu, xy >> 16, n_spp, org, dir);
      // camera rays that miss the (padded) scene bounds cannot hit anything: they are finished right here
      // (radiance = 0 + 1 * environment, pt.cu:504-523) and never enter the traversal queue, so the waves of
      // bounce 0 only hold rays that enter the scene and no path state is written for the others
      RayPre rp;  // (only what the slab test reads: origin and reciprocal direction, as ray_prepare forms them)
      rp.o = org;
      rp.inv = safe_reciprocal(dir);
      float tn;
      enter = alive && slab_test(rp, fr.scene_lo.x, fr.scene_lo.y, fr.scene_lo.z, fr.scene_hi.x, fr.scene_hi.y, fr.scene_hi.z, 1e9f, tn);
      if (enter) {
        pool.ray_o[p] = mk4(org, 1e9f);
        pool.ray_d[p] = mk4(dir, 0.0f);
        pool.thr[p] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        pool.rad[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        pool.pixel[p] = image_idx;
        pool.nspp[p] = n_spp;
        pool.flags[p] = 0u;
      } else {
        const f3 r = alive ? mk3(0.0f) + mk3(1.0f) * env_radiance(fr, dir) : mk3(0.0f);
        pool.rad[p] = mk4(r, 0.0f);
        pool.flags[p] = 2u;  // finished here and in no queue: only k_accumulate reads the slot again (radiance and flags)
      }
    }
    const unsigned long long m = __ballot(enter);
    if ((threadIdx.x & 63u) == 0u) scratch[4 * c + (threadIdx.x >> 6)] = (uint32_t)__popcll(m);
    packed |= ((enter ? 0x80u : 0u) | (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull))) << (8 * c);
   }
   __syncthreads();
   if (threadIdx.x == 0u) {
     uint32_t total = 0;
     for (uint32_t c = 0; c < 4u * kGenChunks; ++c) { const uint32_t v = scratch[c]; scratch[c] = total; total += v; }
     scratch[4 * kGenChunks] = total ? atomicAdd(&pool.counters[CNT_RAD], total) : 0u;
   }
   __syncthreads();
   ++iter;  // (the next round writes the other scratch area: a lane may still be reading this one)
#pragma unroll
   for (int c = 0; c < kGenChunks; ++c)
     if ((packed >> (8 * c)) & 0x80u)
       pool.q_rad[0][scratch[4 * kGenChunks] + scratch[4 * c + (threadIdx.x >> 6)] + ((packed >> (8 * c)) & 63u)] = k * n_owned + base + (uint32_t)c * blockDim.x + threadIdx.x;
  }
}

// samples started per owned pixel, bumped after k_generate has read it for every path of the pass
__global__ void __launch_bounds__(kBlock) k_bump_issued(uint32_t* issued, const uint32_t* owned, uint32_t n_owned, uint32_t n_batch)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_owned) issued[owned[i]] += n_batch;
}

// ------------------------------------------------------------------------------------------------
// Pixels that cannot see the scene.  On the bench's soup 79 % of the pixels are sky from corner to corner: not one ray of their footprint, through any point of
// the lens, reaches the scene's bounds.  Every sample of such a pixel is a camera ray plus an environment lookup; sent through the wavefront machinery it costs a
// path slot of 284 bytes, a radiance record written by k_generate and read back by k_accumulate, and it makes a pass hold a fifth of the paths that do work.  So
// a call that brings enough samples splits its owned pixels once per camera: k_split_pixels sorts them into the pixels the passes render (`wave`) and the sky
// pixels, and k_sky_pixels renders ALL samples of the call for a sky pixel in one go -- camera sample, environment, NaN guard, running means -- with the same
// device functions in the same order as k_generate / k_accumulate, i.e. the same bits, no pool memory and one read and one write of the six layers per call.
//
// The test is conservative.  In camera space every ray of pixel (px, py) starts on the lens disk (centre A0 = (0, 0, f), radius R) and runs through the
// z-mirrored image B of its focus point, B = (k uvx, k uvy, 2 f - (a + b)) with k = 1 - (a + b) / f and (uvx, uvy) inside the pixel, so the rays are the lines
// through a disk of radius R around A0 and a square of half-diagonal rho = sqrt(2) |k| / H around B0.  A point of such a line at parameter s (A + s (B - A))
// is at most R |1 - s| + rho |s| <= R + (R + rho) |s| away from the centre line's point at s; with d0 the distance of the sphere's centre from the centre line,
// tau the position of its foot along it and L = |B0 - A0|, no line of the family comes nearer than d0 sqrt(1 - (m / L)^2) - R - m |tau| / L, m = R + rho.  The
// pixel is sky when that exceeds the bounding sphere of the padded scene bounds with a margin (0.1 % of the radius and of the distance, far above the rounding
// of these few operations).  k_sky_pixels still runs k_generate's own bounds test per sample and counts a violation (fh_sync reports it): the split can only
// lose speed, never a ray that hits.
struct SplitDev {
  f3 centre;      // bounding sphere of the padded scene bounds, camera space
  float radius;
  float far;      // distance of the farthest corner of the padded scene bounds from the lens centre
  uint32_t* wave_px; uint32_t* wave_xy; uint32_t* sky_px; uint32_t* sky_xy;
  uint32_t* counters;  // [0] wave pixels, [1] sky pixels, [2] bounds-test violations seen by k_sky_pixels
};
__global__ void __launch_bounds__(kBlock) k_split_pixels(FrameDev fr, SplitDev sp, const uint32_t* owned, const uint32_t* owned_xy, uint32_t n_owned)
{
  __shared__ uint32_t s_reserve[2][10];
  uint32_t iter = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_owned; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    bool sky = false, wave = false;
    uint32_t px = 0, xy = 0;
    if (i < n_owned) {
      px = owned[i]; xy = owned_xy[i];
      const float x = (float)(xy & 0xffffu) + 0.5f, y = (float)(xy >> 16) + 0.5f;
      const float uvx = -(2.0f * x - (float)fr.width) / (float)fr.height, uvy = (2.0f * y - (float)fr.height) / (float)fr.height;
      const float f = fr.cam_inv_tan, k = 1.0f - fr.cam_a_plus_b / f;
      const f3 a0 = mk3(0.0f, 0.0f, f), b0 = mk3(k * uvx, k * uvy, 2.0f * f - fr.cam_a_plus_b);
      const f3 ab = b0 - a0;
      const float L = length(ab);
      const float m = fabsf(fr.cam_lens_radius) + 1.41421357f * fabsf(k) / (float)fr.height * 1.001f;
      const f3 w = sp.centre - a0;
      const float tau = dot(w, ab) / L;
      const float d0 = sqrt_cr(fmaxf(dot(w, w) - tau * tau, 0.0f));
      const float q = 1.0f - (m / L) * (m / L);
      const float nearest = d0 * sqrt_cr(fmaxf(q, 0.0f)) - fabsf(fr.cam_lens_radius) - m * fabsf(tau) / L;
      sky = L > 0.0f && q > 0.0f && nearest > sp.radius * 1.001f + 1e-3f * (length(w) + 1.0f);  // (NaN anywhere: not sky)
      // The sphere is a loose fit of a box (a cube's has 2.4 times its silhouette): a second test, against the bounds themselves.  A ray of the pixel that reaches a point P
      // of the bounds does so at a parameter |s| <= (|P - A0| + R) / (L - m), and there it is within R + m |s| of the centre line's point at s (above).  So with the bounds
      // grown by d = R + m (far + R) / (L - m) on every side -- `far` the distance of their farthest corner from the lens centre -- the centre LINE, a rigid transform away
      // in world space, passes through the grown box whenever any ray of the pixel reaches the bounds; a centre line that misses the grown box makes the pixel sky.
      if (!sky && L > m * 1.01f) {
        const float d = (fabsf(fr.cam_lens_radius) + m * (sp.far + fabsf(fr.cam_lens_radius)) / (L - m)) * 1.001f + 1e-3f * (sp.far + 1.0f);
        const f3 o = xform_point(fr.cam_xf, a0), dir = xform_dir(fr.cam_xf, (1.0f / L) * ab);  // (unit length: the line parameters below are world distances)
        float tn = -3.0e38f, tf = 3.0e38f;
        bool miss = false;
        const float oo[3] = {o.x, o.y, o.z}, dd[3] = {dir.x, dir.y, dir.z}, lo[3] = {fr.scene_lo.x - d, fr.scene_lo.y - d, fr.scene_lo.z - d}, hi[3] = {fr.scene_hi.x + d, fr.scene_hi.y + d, fr.scene_hi.z + d};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (fabsf(dd[a]) < 1e-12f) miss = miss || oo[a] < lo[a] || oo[a] > hi[a];  // parallel to this pair of planes (a NaN compares false: not a miss)
          else {
            const float t0 = (lo[a] - oo[a]) / dd[a], t1 = (hi[a] - oo[a]) / dd[a];
            tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
          }
        }
        // (the interval is widened by a thousandth of the distances involved: the parameters are quotients of quantities of the scene's size, rounded a few times)
        miss = miss || tn - 1e-3f * (fabsf(tn) + sp.far) > tf + 1e-3f * (fabsf(tf) + sp.far);
        sky = miss && dd[0] == dd[0] && dd[1] == dd[1] && dd[2] == dd[2] && d == d;
      }
      wave = !sky;
    }
    uint32_t* const counters2[2] = {sp.counters, sp.counters + 1};
    const bool act[2] = {wave, sky};
    uint32_t pos[2];
    block_queue_reserve<2>(counters2, act, pos, s_reserve[iter & 1u]);
    ++iter;
    if (wave) { sp.wave_px[pos[0]] = px; sp.wave_xy[pos[0]] = xy; }
    if (sky) { sp.sky_px[pos[1]] = px; sp.sky_xy[pos[1]] = xy; }
  }
}

// all `n_samples` samples of this call for the sky pixels: k_generate's path for a ray that misses the scene bounds and k_accumulate's update of the running means, fused
__global__ void __launch_bounds__(kBlock) k_sky_pixels(FrameDev fr, LayersDev layers, uint32_t* issued, const uint32_t* sky_px, const uint32_t* sky_xy, uint32_t n_sky, uint32_t n_samples,
                                                     uint32_t* violations)
{
  __shared__ SobolRows<1> rows;
  __shared__ HosekSky s_sky;
  stage_sky(fr, s_sky);
  const uint32_t dims[1] = {1u};
  load_sobol_rows<1>(rows, fr.sobol_bytes, dims);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sky; i += gridDim.x * blockDim.x) {
    const uint32_t image_idx = sky_px[i], xy = sky_xy[i];
    const uint32_t first = issued[image_idx];
    uint32_t n_spp = layers.sample_count[image_idx];
    f3 beauty = mk3(layers.beauty[image_idx]), position = mk3(layers.position[image_idx]), normal = mk3(layers.normal[image_idx]), albedo = mk3(layers.albedo[image_idx]);
    float depth = layers.depth[image_idx];
    const float4 tc4 = layers.texcoord[image_idx];
    float tcx = tc4.x, tcy = tc4.y;
    bool violated = false;
    uint32_t blk = 0xffffffffu;  // sixteen consecutive samples share most of their two CMJ draws (fh_sampler.h: cmj_block)
    CmjBlock b0{}, b1{};
    for (uint32_t k = 0; k < n_samples; ++k) {
      f3 org, dir;
      const uint32_t n = first + k;
      if ((n >> 4) != blk) {
        blk = n >> 4;
        b0 = cmj_block(blk, image_idx, 0u, fr.seed_hash);
        b1 = cmj_block(blk, image_idx, 1u, fr.seed_hash);
      }
      const bool alive = camera_ray(fr, rows.m[0], image_idx, xy & 0xffffu, xy >> 16, n, cmj_draw_in_block(b0, n), cmj_draw_in_block(b1, n), org, dir);
      RayPre rp;
      rp.o = org;
      rp.inv = safe_reciprocal(dir);
      float tn;
      violated = violated || (alive && slab_test(rp, fr.scene_lo.x, fr.scene_lo.y, fr.scene_lo.z, fr.scene_hi.x, fr.scene_hi.y, fr.scene_hi.z, 1e9f, tn));
      const f3 L = alive ? mk3(0.0f) + mk3(1.0f) * env_radiance(fr, dir) : mk3(0.0f);  // (k_generate: radiance of a path that ends at the scene bounds)
      const f3 radiance = bad3(L) ? mk3(0.0f) : L;                                      // (k_accumulate: NaN guard, then the running means; a sky sample has no AOVs)
      const float coef = 1.0f / (n_spp + 1.0f);
      const float fn = (float)n_spp;
      beauty = coef * (fn * beauty + radiance);
      position = coef * (fn * position + mk3(0.0f));
      normal = coef * (fn * normal + mk3(0.0f));
      depth = coef * (fn * depth + 0.0f);
      tcx = coef * (fn * tcx + 0.0f);
      tcy = coef * (fn * tcy + 0.0f);
      albedo = coef * (fn * albedo + mk3(0.0f));
      n_spp++;
    }
    if (violated) atomicAdd(violations, 1u);
    issued[image_idx] = first + n_samples;
    layers.sample_count[image_idx] = n_spp;
    layers.beauty[image_idx] = mk4(beauty, 1.0f);
    layers.position[image_idx] = mk4(position, 1.0f);
    layers.normal[image_idx] = mk4(normal, 1.0f);
    layers.depth[image_idx] = depth;
    layers.texcoord[image_idx] = make_float4(tcx, tcy, 0.0f, 1.0f);
    layers.albedo[image_idx] = mk4(albedo, 1.0f);
  }
}

// ------------------------------------------------------------------------------------------------
// closest hit: one lane per ray, grid-stride over the queue (binary-BVH fallback shares this kernel)
template <bool COUNT, bool WIDE, bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_closest_static(SceneDev sc, PoolDev pool, uint32_t depth, TraceCounters tc)
{
  const uint32_t* cnt = pool.counters + depth * kCounterStride;
  const uint32_t count = cnt[CNT_RAD];
  const uint32_t* q = pool.q_rad[depth & 1u];
  uint32_t nn = 0, nt = 0, nr = 0;
  WaveSteps ws;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint32_t p = q[i];
    if (COUNT) nr++;
    const float4 o = pool.ray_o[p], d = pool.ray_d[p];
    HitRec h;
    const uint32_t nn0 = nn;
    if (WIDE) traverse_bvh8<false, COUNT, false, ALPHA>(sc.bvh8, mk3(o), mk3(d), o.w, h, nn, nt, &ws, nullptr, 0, &sc);
    else traverse_bvh2<false, COUNT, ALPHA>(sc.bvh2, mk3(o), mk3(d), o.w, h, nn, nt, &sc);
    if (COUNT) { const uint32_t k = nn - nn0; int b = 0; while (b < 7 && k > (8u << b)) ++b; atomicAdd(tc.hist + b, 1ull); }
    pool.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
    pool.q_prim[i] = h.prim;
  }
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)nr);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// closest hit over the 8-wide BVH, wave-cooperative triangle tests (fh_trace.h: traverse_bvh8_coop)
template <bool COUNT, bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_closest_coop(SceneDev sc, PoolDev pool, uint32_t depth, TraceCounters tc, uint32_t flush)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // [entry][thread], sized by the launcher for the depth of the BVH
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kBlock / 64) * kCoopLdsBytesPerWave];
  const CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  const uint32_t* cnt = pool.counters + depth * kCounterStride;
  const uint32_t count = cnt[CNT_RAD];
  const uint32_t* q = pool.q_rad[depth & 1u];
  uint32_t nn = 0, nt = 0, nr = 0;
  WaveSteps ws;
  // the loop bound is per wave, so that all 64 lanes enter the traversal together
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + (threadIdx.x & 63u);
    const bool valid = i < count;
    const uint32_t p = valid ? q[i] : 0u;
    if (COUNT && valid) nr++;
    const float4 o = valid ? pool.ray_o[p] : make_float4(0.0f, 0.0f, 0.0f, 0.0f), d = valid ? pool.ray_d[p] : make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    HitRec h;
    const uint32_t nn0 = nn;
    traverse_bvh8_coop<false, COUNT, true, ALPHA>(sc.bvh8, valid, mk3(o), mk3(d), o.w, h, nn, nt, &ws, cl, flush, lds_stack, (int)sc.bvh8.depth, &sc);
    if (COUNT && valid) { const uint32_t k = nn - nn0; int b = 0; while (b < 7 && k > (8u << b)) ++b; atomicAdd(tc.hist + b, 1ull); }
    if (valid) { pool.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim)); pool.q_prim[i] = h.prim; }
  }
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)nr);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming kernels (fh_trace.h: traverse_stream): the grid is sized to the resident wave slots of the chip; every wave draws
// chunks of the bounce's queue from a global cursor (ChunkFeed) and hands the rays to its lanes as they become idle.

// instrumented build: per-lane histogram of nodes visited per ray (8 bins x 16 bits; a lane traces far fewer than 65535 rays),
// flushed once per lane instead of one contended atomic per ray
struct HistPack {
  unsigned long long w[2] = {0ull, 0ull};
  FH_D void add(uint32_t nodes) { int b = 0; while (b < 7 && nodes > (8u << b)) ++b; if (b < 4) w[0] += 1ull << (16 * b); else w[1] += 1ull << (16 * (b - 4)); }
  FH_D void flush(unsigned long long* hist) const
  {
    for (int b = 0; b < 8; ++b) { const unsigned long long v = (w[b >> 2] >> (16 * (b & 3))) & 0xffffull; if (v) atomicAdd(hist + b, v); }
  }
};

// Queue entries a wave takes per global atomic.  `chunk` packs the range the launcher allows: low 16 bits the default, high 16 bits the maximum.  The cursor
// is ONE address and serves ~55 M returning atomics/s: at 64 rays per atomic that caps a launch at ~3.5 G rays/s, which rays through a tiny BVH exceed
// (Cornell box: 4.2 -> 13 G closest-hit rays/s with 256).  Large chunks cost elsewhere -- a queue shorter than waves x chunk leaves waves without work, and on a
// big BVH consecutive chunks traced by different waves at the same time share the caches -- so the launcher raises the maximum only for small trees and the
// kernel stays below a quarter of a wave's even share of the queue.
FH_D uint32_t stream_chunk_for(uint32_t count, uint32_t chunk)
{
  const uint32_t lo = chunk & 0xffffu, hi = chunk >> 16;
  if (hi <= lo) return lo;
  const uint32_t share = count / (gridDim.x * (kBlock / 64u) * 4u);
  const uint32_t c = share & ~63u;
  return c < lo ? lo : (c > hi ? hi : c);
}

// Small launches (the reference's callers render 1 or 16 samples per call, controller.cpp:224, rtcamp8.cpp:183-189).  The grid of a streaming launch is every
// resident wave slot of the chip; with a short queue every one of those waves draws one chunk and runs it at a fraction of its lanes, six to a SIMD, all
// competing for the SIMD's issue slots: the launch then takes (steps of its longest ray) x (six mostly empty waves' node tests) -- 0.34 ms for the 330 k
// camera rays of a 1-spp 1080p frame of the 1 M-triangle scene, which at the kernel's throughput are 0.07 ms of work.  So only as many workgroups take part
// as the queue can give `min_rays` entries per wave: the others leave at once, the SIMDs hold one or two waves whose steps are as short as memory latency lets them
// be, and in-wave refill keeps their lanes busy.  Big launches are untouched (every workgroup takes part from 6144 x min_rays entries on).
FH_D bool stream_block_idle(uint32_t count, uint32_t min_rays) { return blockIdx.x != 0u && (unsigned long long)blockIdx.x * (kBlock / 64u) * min_rays >= count; }

template <bool ALPHA>
struct AlphaLds { static FH_D void attach(CoopLds&) {} };
template <>
struct AlphaLds<true> {
  static FH_D void attach(CoopLds& cl)
  {
    __shared__ __attribute__((aligned(16))) unsigned char lds_alpha[kAlphaLdsBytesPerBlock];
    alpha_ring(cl, lds_alpha, threadIdx.x >> 6);
  }
};

// direction.w of a ray record (store_secondary below, PoolSink::next): 0 = no ray; otherwise 1 + the wide node the ray starts its traversal at (1: the root)
FH_D bool sec_present(const float4& d4) { return __float_as_uint(d4.w) != 0u; }
FH_D uint32_t start_node_of(const float4& d4) { const uint32_t b = __float_as_uint(d4.w); return b ? b - 1u : 0u; }

template <bool COUNT>
struct ClosestStream {
  static constexpr bool all_any = false;
  static constexpr bool can_climb = false;
  const PoolDev& pool;
  const uint32_t* q;
  ChunkFeed feed;
  uint32_t p = 0, qi = 0;  // path slot and queue entry of the lane's ray
  uint32_t start = 0;      // wide node the ray starts its traversal at (0: the root; camera rays)
  uint32_t n_rays = 0;
  unsigned long long* hist;
  HistPack hp;
  FH_D ClosestStream(const PoolDev& pl, const uint32_t* qq, const ChunkFeed& f, unsigned long long* hs) : pool(pl), q(qq), feed(f), hist(hs) {}
  FH_D bool advance(f3&, f3&, float&, bool&) { return false; }
  FH_D bool take(uint32_t i, f3& o, f3& d, float& tmax, bool& any)
  {
    if (i >= feed.end) return false;
    p = q[i];
    qi = i;
    const float4 o4 = pool.ray_o[p], d4 = pool.ray_d[p];
    o = mk3(o4); d = mk3(d4); tmax = o4.w; any = false;
    start = start_node_of(d4);
    if (COUNT) n_rays++;
    return true;
  }
  FH_D void commit(bool, const HitRec& h, uint32_t nodes)
  {
    pool.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
    pool.q_prim[qi] = h.prim;
    if (COUNT) hp.add(nodes);
  }
  FH_D bool drained() const { return feed.drained(); }
  FH_D bool followup() const { return false; }
  FH_D uint32_t start_node() const { return start; }
};

template <bool COUNT, bool ALPHA>
__global__ void __launch_bounds__(kBlock, COUNT ? 1 : FH_STREAM_BLOCKS_CLOSEST) k_trace_closest_stream(SceneDev sc, PoolDev pool, uint32_t depth, TraceCounters tc, uint32_t flush, uint32_t refill, uint32_t chunk, uint32_t min_rays,
                                                                 StackSpill spill)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kBlock / 64) * kCoopLdsBytesPerWave];
  const uint32_t count = pool.counters[depth * kCounterStride + CNT_RAD];
  if (stream_block_idle(count, min_rays)) return;
  const ClockStamp stamp;
  CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  AlphaLds<AlphaDefer<false, ALPHA>::value>::attach(cl);  // candidates waiting for their any-hit test (fh_trace.h: alpha_ring): LDS of the kernels with the test compiled in only
  uint32_t nn = 0, nt = 0;
  WaveSteps ws;
  ClosestStream<COUNT> pol(pool, pool.q_rad[depth & 1u], ChunkFeed(pool.counters + depth * kCounterStride + CNT_CUR_CLOSEST, count, stream_chunk_for(count, chunk)), tc.hist);
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // the traversal stack of every lane: [entry][thread], as many entries as the BVH has levels
  // (no LDS copy of the top nodes here, fh_trace.h stage_top_nodes: the closest-hit launch gained 0.9 % alone on configs[2] and LOST 1.7 % on configs[3], where the ring of parked any-hit tests already takes its LDS)
  traverse_stream<false, COUNT, true, ALPHA>(sc.bvh8, pol, nn, nt, &ws, cl, flush, refill, lds_stack, (int)sc.bvh8.depth, &sc, spill);
  stamp.commit(tc.clk);
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)pol.n_rays);
    pol.hp.flush(tc.hist);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// ------------------------------------------------------------------------------------------------
// Sort the hits of this bounce into per-class queues.  Appends are aggregated per block: wave
// ballot + popcount, a prefix over the block's waves in LDS, one atomic per block and class.
constexpr int kRouteChunks = 8;
__global__ void __launch_bounds__(kBlock) k_route(SceneDev sc, PoolDev pool, uint32_t depth, uint32_t n_classes, unsigned long long* hit_counter)
{
  // A workgroup takes kRouteChunks x 256 consecutive entries per round and reserves its part of every class queue with ONE returning atomic per class: a class counter is one
  // address, the chip serves ~85 M returning atomics per address and second, and with 256 entries per atomic the 357 M closest hits of a configs[2] frame were 1.4 M atomics per
  // counter -- 16.4 ms of the kernel's 16.6 (profiles/README.md r4-20)
  __shared__ uint32_t wave_cnt[kMaxClasses][kRouteChunks * (kBlock / 64)];
  __shared__ uint32_t block_base[kMaxClasses];
  uint32_t* cnt = pool.counters + depth * kCounterStride;
  const uint32_t count = cnt[CNT_RAD];
  const uint32_t* q = pool.q_rad[depth & 1u];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t base = blockIdx.x * blockDim.x * kRouteChunks; base < count; base += gridDim.x * blockDim.x * kRouteChunks) {
    uint32_t p[kRouteChunks], cls[kRouteChunks], rank[kRouteChunks];
#pragma unroll
    for (int c = 0; c < kRouteChunks; ++c) {
      const uint32_t i = base + (uint32_t)c * blockDim.x + threadIdx.x;
      p[c] = 0u; cls[c] = 0xffu; rank[c] = 0u;
      if (i < count) {
        p[c] = q[i];
        const uint32_t prim = pool.q_prim[i];  // (= the face id bits of pool.hit[p].w, from a stream)
        if (prim != 0xffffffffu) cls[c] = sc.face_cls[prim] & 0x1fu;
      }
    }
#pragma unroll
    for (int c = 0; c < kRouteChunks; ++c)
      for (uint32_t k = 0; k < n_classes; ++k) {
        const unsigned long long m = __ballot(cls[c] == k);
        if (lane == 0) wave_cnt[k][c * (kBlock / 64) + wave] = (uint32_t)__popcll(m);
        if (cls[c] == k) rank[c] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      }
    __syncthreads();
    if (threadIdx.x < n_classes) {
      uint32_t total = 0;
      for (uint32_t w = 0; w < kRouteChunks * (kBlock / 64); ++w) { const uint32_t v = wave_cnt[threadIdx.x][w]; wave_cnt[threadIdx.x][w] = total; total += v; }
      block_base[threadIdx.x] = total ? atomicAdd(&cnt[CNT_CLS + threadIdx.x], total) : 0u;
      if (hit_counter && total) atomicAdd(hit_counter, (unsigned long long)total);  // instrumented runs: surface hits that get shaded
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kRouteChunks; ++c)
      if (cls[c] < n_classes) pool.q_cls[(size_t)cls[c] * pool.capacity + block_base[cls[c]] + wave_cnt[cls[c]][c * (kBlock / 64) + wave] + rank[c]] = p[c];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// paths that leave the scene at depth 0 see the environment directly (pt.cu:504-523)
__global__ void __launch_bounds__(kBlock) k_miss_primary(FrameDev fr, PoolDev pool)
{
  __shared__ HosekSky s_sky;
  stage_sky(fr, s_sky);
  __syncthreads();
  const uint32_t count = pool.counters[CNT_RAD];
  const uint32_t* q = pool.q_rad[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    if (pool.q_prim[i] != 0xffffffffu) continue;  // (the closest hit's face id of queue entry i, as k_route reads it: no path record is touched for a ray that hit)
    const uint32_t p = q[i];
    if (pool.flags[p] & 4u) continue;  // bug-compat mode: an earlier sample of this launch already hit something (k_firsthit_scan)
    const f3 T = mk3(pool.thr[p]);
    const f3 d = mk3(pool.ray_d[p]);
    const f3 r = mk3(pool.rad[p]) + T * env_radiance(fr, d);
    pool.rad[p] = mk4(r, 0.0f);
  }
}

// ------------------------------------------------------------------------------------------------
FH_D int tex_id(const MaterialDev& m, int word) { return __float_as_int(m.w[word]); }

// fill_shading_params (pt.cu:181-280): material constants, overridden by texture lookups where a texture id is set
FH_D MatParams load_params(const SceneDev& sc, const MaterialDev& m, float tu, float tv)
{
  MatParams p;
  p.diffuse = m.w[0];
  p.base_color = tex_id(m, 4) >= 0 ? tex_rgb(sc, tex_id(m, 4), tu, tv) : mk3(m.w[1], m.w[2], m.w[3]);
  p.diffuse_roughness = m.w[5];
  p.specular = m.w[6];
  p.specular_color = tex_id(m, 10) >= 0 ? tex_rgb(sc, tex_id(m, 10), tu, tv) : mk3(m.w[7], m.w[8], m.w[9]);
  p.specular_roughness = clampf(tex_id(m, 12) >= 0 ? tex_rgba(sc, tex_id(m, 12), tu, tv).x : m.w[11], 0.01f, 1.0f);
  p.metalness = tex_id(m, 14) >= 0 ? tex_rgba(sc, tex_id(m, 14), tu, tv).x : m.w[13];
  if (tex_id(m, 15) >= 0) {  // glTF metallic-roughness texture: G = roughness, B = metalness (pt.cu:230-236)
    const float4 mr = tex_rgba(sc, tex_id(m, 15), tu, tv);
    p.specular_roughness = clampf(mr.y, 0.01f, 1.0f);
    p.metalness = clampf(mr.z, 0.0f, 1.0f);
  }
  p.coat = clampf(tex_id(m, 17) >= 0 ? tex_rgba(sc, tex_id(m, 17), tu, tv).x : m.w[16], 0.0f, 1.0f);
  p.coat_color = mk3(1.0f, 1.0f, 1.0f);  // fill_shading_params never copies material.coat_color (pt.cu:238-255)
  p.coat_roughness = clampf(tex_id(m, 22) >= 0 ? tex_rgba(sc, tex_id(m, 22), tu, tv).y : m.w[21], 0.0f, 1.0f);
  p.transmission = m.w[23];
  p.transmission_color = mk3(m.w[24], m.w[25], m.w[26]);
  p.sheen = m.w[27];
  p.sheen_color = mk3(m.w[28], m.w[29], m.w[30]);
  p.sheen_roughness = m.w[31];
  p.subsurface = m.w[32];
  p.subsurface_color = mk3(m.w[33], m.w[34], m.w[35]);
  p.thin_walled = m.w[36];
  return p;
}

// get_emission (pt.cu:131-139)
FH_D f3 emission_of(const SceneDev& sc, const MaterialDev& m, float tu, float tv)
{
  return tex_id(m, 41) >= 0 ? tex_rgb(sc, tex_id(m, 41), tu, tv) : mk3(m.w[38], m.w[39], m.w[40]);
}

// (direction.w of a ray record: 0 = no ray in this place; otherwise 1 + the wide node the ray starts its traversal at -- the node that holds the face it leaves, fh_trace.h:
// bottom-up start -- or 1 = the root.  Read as BITS everywhere: small integers are denormal floats.)
FH_D void store_secondary(const PoolDev& pool, uint32_t slot, uint32_t p, f3 o, float tmax, f3 d, bool active, f3 c, uint32_t start_bits = 1u)
{
  const size_t k = pool.sec_at(slot, p);
  pool.sec[k + 1] = mk4(d, __uint_as_float(active ? start_bits : 0u));
  if (!active) return;  // (nobody reads origin or contribution of a place without a ray)
  pool.sec[k] = mk4(o, tmax);
  pool.sec[k + 2] = mk4(c, 0.0f);
}


// Sobol' dimensions and CMJ slots of one bounce
struct BounceSlots {
  uint32_t has_lights;
  uint32_t dim_area, dim_light, dim_next, dim_rr;
  uint32_t slot_dir, slot_sky, slot_area, slot_light, slot_next;
  FH_D void set(const FrameDev& fr, uint32_t n_lights, uint32_t depth)
  {
    has_lights = n_lights > 0 ? 1u : 0u;
    const uint32_t dim0 = 1u + depth * fr.n1;  // this bounce's RR dimension (already consumed)
    dim_area = dim0 + 1u;                      // only drawn if has_lights
    dim_light = dim0 + 1u + has_lights;
    dim_next = dim_light + 1u;
    dim_rr = 1u + (depth + 1u) * fr.n1;
    const uint32_t slot0 = 2u + depth * fr.n2;
    slot_dir = slot0;
    slot_sky = slot0 + fr.has_dir;
    slot_area = slot_sky + 1u;
    slot_light = slot_sky + 1u + has_lights;
    slot_next = slot_light + 1u;
  }
  FH_D void load_rows(SobolRows<4>& rows, const uint32_t* tables) const
  {
    const uint32_t dims[4] = {dim_area, dim_light, dim_next, dim_rr};
    load_sobol_rows<4>(rows, tables, dims);
  }
};

struct SecRay { f3 o; float tmax; f3 d; bool active; f3 c; };

// A shadow ray whose pre-multiplied contribution is exactly (+-0, +-0, +-0) cannot change a bit of the result whatever it hits: the radiance it would be added to starts at
// +0 and a sum is -0 only when both terms are, so x + (+-0) == x for every radiance x that can occur.  The reference traces it all the same (pt.cu:837-857 sends the sky's
// shadow ray for a black constant background too: one of three secondary rays per bounce of a Cornell box); here such a ray is stored as "no ray in this
// place".  NaN compares unequal to 0 and stays: a NaN contribution of an unoccluded ray must still reach the radiance and zero the sample (pt.cu:474-478).
FH_D bool contributes(f3 c) { return !(c.x == 0.0f && c.y == 0.0f && c.z == 0.0f); }

// What one shaded hit produces (pt.cu:680-944) goes to a SINK as soon as it exists, so that a kernel that only stores the products
// (k_shade) does not keep them alive in registers next to the BSDF state; k_tail, which traces the rays itself, collects them in
// registers.  Sink interface:
//   aov(position, normal, albedo, u, v)         first hit only                              (pt.cu:745-751)
//   emissive(L)                                 first hit on an emitter: the path ends      (pt.cu:754-759)
//   secondary(slot, origin, tmax, dir, active, contribution)      one per enabled NEE slot, in the reference's order
//   light_pending(T, cos, f, pdf)               scenes with emitters: what the light ray's MIS weight needs once its hit is known
//   next(origin, dir, T)                        the path continues
struct ShadeOut {      // register sink (k_tail)
  bool emissive_done = false;  // first hit on an emitter: radiance updated, path ends, nothing else valid
  bool shaded = false;         // secondary rays valid
  bool cont = false;           // next ray valid
  f3 L;                // radiance after a directly visible emitter
  f3 T;                // throughput for the next bounce (after Russian roulette)
  SecRay sec[SEC_COUNT];
  f3 lp_T, lp_f;       // light ray with emitters: throughput before the update, BSDF value
  float lp_cos, lp_pdf;
  f3 next_o, next_d;
  FH_D ShadeOut() { for (uint32_t k = 0; k < SEC_COUNT; ++k) sec[k].active = false; }
  FH_D void aov(f3, f3, f3, float, float) {}
  FH_D void emissive(f3 l) { L = l; emissive_done = true; }
  FH_D void secondary(uint32_t slot, f3 o, float tmax, f3 d, bool active, f3 c) { shaded = true; sec[slot] = SecRay{o, tmax, d, active, c}; }
  FH_D void light_pending(f3 t, float c, f3 f, float pdf) { lp_T = t; lp_cos = c; lp_f = f; lp_pdf = pdf; }
  FH_D void next(f3 o, f3 d, f3 t) { cont = true; next_o = o; next_d = d; T = t; }
  FH_D void start_node(uint32_t) {}  // (the fused tail's rays start at the root)
};

struct PoolSink {      // memory sink (k_shade): path slot p of the pool
  const PoolDev& pool;
  uint32_t p;
  float hit_t;
  bool shaded = false, cont = false;
  f3 origin;           // where the path's rays leave the surface (cell key of the bounce queues)
  uint32_t start_bits; // 1 + the wide node that holds the face the rays leave (1: they start at the root)
  FH_D PoolSink(const PoolDev& pl, uint32_t slot, float t, uint32_t sb = 1u) : pool(pl), p(slot), hit_t(t), origin(mk3(0.0f)), start_bits(sb) {}
  FH_D void aov(f3 position, f3 normal, f3 albedo, float u, float v)
  {
    pool.aov_position[p] = mk4(position, 0.0f);
    pool.aov_normal[p] = mk4(normal, 0.0f);
    pool.aov_albedo[p] = mk4(albedo, 0.0f);
    pool.aov_texdepth[p] = make_float4(u, v, hit_t, 0.0f);
    pool.flags[p] |= 1u;
  }
  FH_D void emissive(f3 l) { pool.rad[p] = mk4(l, 0.0f); }
  FH_D void start_node(uint32_t bits) { start_bits = bits ? bits : 1u; }
  FH_D void secondary(uint32_t slot, f3 o, float tmax, f3 d, bool active, f3 c)
  {
    store_secondary(pool, slot, p, o, tmax, d, active, c, start_bits);
    if (!shaded) origin = o;  // all rays of a path leave (almost) the same point
    shaded = true;
  }
  FH_D void light_pending(f3 t, float c, f3 f, float pdf) { pool.lp_a[p] = mk4(t, c); pool.lp_b[p] = mk4(f, pdf); }
  FH_D void next(f3 o, f3 d, f3 t)
  {
    pool.ray_o[p] = mk4(o, 1e9f);
    pool.ray_d[p] = mk4(d, __uint_as_float(start_bits));
    pool.thr[p] = mk4(t, 0.0f);
    if (!shaded) origin = o;
    cont = true;
  }
};

template <uint32_t LOBES, class Sink>
FH_D void shade_hit(const SceneDev& sc, const FrameDev& fr, const SobolRows<4>& rows, const BounceSlots& bs, uint32_t depth, float4 hit, f3 rd, f3 T, f3 L, uint32_t image_idx, uint32_t n_spp,
                    Sink& out, bool first = true)
{
  const uint32_t has_lights = bs.has_lights;
  const uint32_t prim = __float_as_uint(hit.w);
  const float bu = hit.y, bv = hit.z;
  const uint32_t sidx = image_idx + n_spp * fr.width * fr.height;

  // surface (pt.cu:141-179) from the pre-transformed face record
  const float4 r0 = sc.face_rec[kFaceRec * (size_t)prim], r1 = sc.face_rec[kFaceRec * (size_t)prim + 1], r2 = sc.face_rec[kFaceRec * (size_t)prim + 2];
  const float4 r3 = sc.face_rec[kFaceRec * (size_t)prim + 3], r4 = sc.face_rec[kFaceRec * (size_t)prim + 4], r5 = sc.face_rec[kFaceRec * (size_t)prim + 5];
  const float4 r6 = sc.face_rec[kFaceRec * (size_t)prim + 6];
  // (.z of the record's last vector: 1 + the wide node that holds the face, written by the BVH build -- where this bounce's first-hit rays start their traversal, fh_trace.h)
  out.start_node(sc.face_node ? __float_as_uint(r6.z) : 0u);
  const f3 p0 = mk3(r0), p1 = mk3(r1), p2 = mk3(r2);
  const float bw = 1.0f - bu - bv;
  const f3 x = bw * p0 + bu * p1 + bv * p2;
  f3 ng = normalize(cross(p1 - p0, p2 - p0));
  f3 ns = normalize(bw * mk3(r3) + bu * mk3(r4) + bv * mk3(r5));
  const float tu = bw * r0.w + bu * r2.w + bv * r4.w;
  const float tv = bw * r1.w + bu * r3.w + bv * r5.w;
  const bool entering = dot(-rd, ng) > 0;
  ns = entering ? ns : -ns;
  ng = entering ? ng : -ng;
  f3 tangent, bitangent;
  onb(ns, tangent, bitangent);
  const MaterialDev& mat = sc.materials[__float_as_uint(r6.x)];
  const MatParams sp = load_params(sc, mat, tu, tv);
  const f3 ns_geo = ns, tangent_geo = tangent, bitangent_geo = bitangent;  // surf_info frame before the maps
  if (tex_id(mat, 42) >= 0) {  // bump mapping with a height map (pt.cu:709-731)
    const fht_texture& hm = sc.textures[tex_id(mat, 42)];
    const float du = 1.0f / hm.width, dv = 1.0f / hm.height;
    const float hv = tex_rgba(sc, tex_id(mat, 42), tu, tv).x;
    const float dfdu = tex_rgba(sc, tex_id(mat, 42), tu + du, tv).x - hv;
    const float dfdv = tex_rgba(sc, tex_id(mat, 42), tu, tv + dv).x - hv;
    tangent = normalize(tangent_geo + dfdu * ns_geo);
    bitangent = normalize(bitangent_geo + dfdv * ns_geo);
    ns = normalize(cross(tangent, bitangent));
  }
  if (tex_id(mat, 43) >= 0) {  // normal mapping (pt.cu:733-742)
    f3 value = tex_rgb(sc, tex_id(mat, 43), tu, tv);
    value = 2.0f * value - 1.0f;
    ns = normalize(to_world(value, tangent_geo, bitangent_geo, ns_geo));
    onb(ns, tangent, bitangent);
  }

  if (depth == 0 && first) {  // first hit: AOVs and directly visible emitters (pt.cu:745-760); `first` is false only in the bug-compat mode (k_firsthit_scan)
    out.aov(x, ns, sp.base_color, tu, tv);
    if (mat.emissive) {
      out.emissive(L + T * emission_of(sc, mat, tu, tv));
      return;
    }
  }
  const f3 wo = to_local(-rd, tangent, ns, bitangent);
  Bsdf<LOBES> bsdf;
  bsdf.init(wo, sp, entering, fr.lut);
  const f3 so = offset_origin(x, ng);

  // directional light (pt.cu:772-793)
  if (fr.has_dir) {
    const f2 pdisk = concentric_disk(cmj_draw(n_spp, image_idx, bs.slot_dir, fr.seed_hash));
    f3 t, b;
    onb(fr.dir_dir, t, b);
    const f3 pl = 1e9f * fr.dir_dir + fr.dir_disk_radius * (t * pdisk.x + b * pdisk.y);
    const f3 sd = normalize(pl - so);
    const f3 wi = to_local(sd, tangent, ns, bitangent);
    const f3 f = bsdf.eval(wo, wi);
    const float pdf = 1.0f;
    const float w = pdf / (pdf + bsdf.eval_pdf(wo, wi));
    const f3 c = clamp01(T * w * f * abs_cos(wi) / pdf) * fr.dir_le;
    out.secondary(SEC_DIR, so, 1e9f - 0.001f, sd, contributes(c), c);
  }
  // sky / constant background (pt.cu:817-857)
  {
    const f3 wi = cosine_hemisphere(cmj_draw(n_spp, image_idx, bs.slot_sky, fr.seed_hash));
    const f3 sd = to_world(wi, tangent, ns, bitangent);
    const f3 f = bsdf.eval(wo, wi);
    const float pdf = abs_cos(wi) / kPi;
    const float w = pdf / (pdf + bsdf.eval_pdf(wo, wi));
    const f3 c = clamp01(T * w * f * abs_cos(wi) / pdf) * env_radiance(fr, sd);
    out.secondary(SEC_SKY, so, 1e9f - 0.001f, sd, contributes(c), c);
  }
  // area lights (pt.cu:860-889, :282-322)
  if (has_lights) {
    const float u1 = sobol_draw_bytes(rows.m[0], sidx, bs.dim_area, fr.seed_hash);
    const f2 u2 = cmj_draw(n_spp, image_idx, bs.slot_area, fr.seed_hash);
    uint32_t li = (uint32_t)(u1 * sc.n_lights);
    li = li < sc.n_lights - 1u ? li : sc.n_lights - 1u;
    const AreaLightDev lt = sc.lights[li];
    const f2 bc = triangle_barycentric(u2);
    const float4 l0 = sc.face_rec[kFaceRec * (size_t)lt.face], l1 = sc.face_rec[kFaceRec * (size_t)lt.face + 1], l2 = sc.face_rec[kFaceRec * (size_t)lt.face + 2];
    const float4 l3 = sc.face_rec[kFaceRec * (size_t)lt.face + 3], l4 = sc.face_rec[kFaceRec * (size_t)lt.face + 4], l5 = sc.face_rec[kFaceRec * (size_t)lt.face + 5];
    const float lw = 1.0f - bc.x - bc.y;
    const f3 lp = lw * mk3(l0) + bc.x * mk3(l1) + bc.y * mk3(l2);
    const f3 ln = lw * mk3(l3) + bc.x * mk3(l4) + bc.y * mk3(l5);
    const float area = 0.5f * length(cross(mk3(l1) - mk3(l0), mk3(l2) - mk3(l0)));
    const MaterialDev& lm = sc.materials[lt.material];
    const f3 le = emission_of(sc, lm, lw * l0.w + bc.x * l2.w + bc.y * l4.w, lw * l1.w + bc.x * l3.w + bc.y * l5.w);
    const float pdf_area = 1.0f / (sc.n_lights * area);
    const f3 sd = normalize(lp - so);
    const float r = length(lp - so);
    const bool facing = dot(-sd, ln) > 0.0f;
    const f3 wi = to_local(sd, tangent, ns, bitangent);
    const f3 f = bsdf.eval(wo, wi);
    const float pdf = r * r / fabsf(dot(-sd, ln)) * pdf_area;
    const float w = pdf / (pdf + bsdf.eval_pdf(wo, wi));
    const f3 c = clamp01(T * w * f * abs_cos(wi) / pdf) * le;
    out.secondary(SEC_AREA, so, r - 0.001f, sd, facing && contributes(c), c);
  }
  // BSDF-sampled light ray (pt.cu:893-925)
  {
    f3 f;
    float pdf;
    const float u1 = sobol_draw_bytes(rows.m[1], sidx, bs.dim_light, fr.seed_hash);
    const f2 u2 = cmj_draw(n_spp, image_idx, bs.slot_light, fr.seed_hash);
    const f3 wi = bsdf.sample(wo, u1, u2, f, pdf);
    const f3 ld = to_world(wi, tangent, ns, bitangent);
    const bool transmitted = dot(ld, ng) < 0;
    const f3 lo = offset_origin(x, transmitted ? -ng : ng);
    if (has_lights) {
      // the MIS weight needs the hit (emitter or sky): finished once the closest hit is known
      out.light_pending(T, abs_cos(wi), f, pdf);
      out.secondary(SEC_LIGHT, lo, 1e9f, ld, true, mk3(0.0f));
    } else {
      // no emitters: the ray contributes only if it escapes, with the sky's cosine pdf (pt.cu:917-919)
      const float pdf_light = abs_cos(wi) / kPi;
      const float w = pdf / (pdf + pdf_light);
      const f3 c = clamp01(T * w * f * abs_cos(wi) / pdf) * env_radiance(fr, ld);
      out.secondary(SEC_LIGHT, lo, 1e9f, ld, contributes(c), c);
    }
  }
  // next direction (pt.cu:928-943) and the next bounce's Russian roulette (pt.cu:457-471)
  {
    f3 f;
    float pdf;
    const float u1 = sobol_draw_bytes(rows.m[2], sidx, bs.dim_next, fr.seed_hash);
    const f2 u2 = cmj_draw(n_spp, image_idx, bs.slot_next, fr.seed_hash);
    const f3 wi = bsdf.sample(wo, u1, u2, f, pdf);
    const f3 wd = to_world(wi, tangent, ns, bitangent);
    f3 Tn = T;
    Tn *= f * abs_cos(wi) / pdf;
    const bool transmitted = dot(wd, ng) < 0;
    const f3 no = offset_origin(x, transmitted ? -ng : ng);
    if (!bad3(Tn) && depth + 1u < fr.max_depth) {
      const float prr = clampf(lum(Tn), 0.0f, 1.0f);
      const float u = sobol_draw_bytes(rows.m[3], sidx, bs.dim_rr, fr.seed_hash);
      if (!(u >= prr)) out.next(no, wd, Tn / prr);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Spatial ordering of the bounce queues.  Rays leaving hit points are incoherent; traced in the order the shade kernels emit them,
// the lanes of a wave walk unrelated parts of the BVH and every L2 sees all of it.  Sorting 4 M random rays of the 1 M-triangle
// soup by the Morton cell of their origin made the same traversal kernel 1.4-1.55x faster (tools/sort_probe.py), so the queue
// entries are brought into cell order first: a counting sort over kCells keys (histogram, scan, scatter), ~0.3 ms for 40 M entries.
// Only the ORDER of queue entries changes: every path's arithmetic is untouched, so results are bit-identical.
FH_D uint32_t spread3(uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }  // 4 bits -> every third bit
FH_D uint32_t cell_of(const FrameDev& fr, f3 p)
{
  const float m = (float)((1u << kCellBits) - 1u);
  const float fx = fminf(fmaxf((p.x - fr.scene_lo.x) * fr.cell_scale.x, 0.0f), m), fy = fminf(fmaxf((p.y - fr.scene_lo.y) * fr.cell_scale.y, 0.0f), m),
              fz = fminf(fmaxf((p.z - fr.scene_lo.z) * fr.cell_scale.z, 0.0f), m);
  return spread3((uint32_t)fx) | (spread3((uint32_t)fy) << 1) | (spread3((uint32_t)fz) << 2);
}

constexpr int kSortBlock = 1024;
// contiguous share of block b of n items
FH_D void block_share(uint32_t n, uint32_t& lo, uint32_t& hi)
{
  const uint32_t per = (n + gridDim.x - 1u) / gridDim.x;
  lo = blockIdx.x * per < n ? blockIdx.x * per : n;
  hi = lo + per < n ? lo + per : n;
}

__global__ void __launch_bounds__(kSortBlock) k_cell_hist(const uint32_t* count_ptr, const uint16_t* keys, uint32_t* hist)
{
  __shared__ uint32_t h[kCells];
  for (uint32_t b = threadIdx.x; b < kCells; b += kSortBlock) h[b] = 0u;
  __syncthreads();
  uint32_t lo, hi;
  block_share(*count_ptr, lo, hi);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kSortBlock) atomicAdd(&h[keys[i] & (kCells - 1u)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < kCells; b += kSortBlock)
    if (h[b]) atomicAdd(&hist[b], h[b]);
}

// exclusive scan of the histogram into the cursors; the histogram is left zero for the next sort
__global__ void __launch_bounds__(kSortBlock) k_cell_scan(uint32_t* hist, uint32_t* cursor)
{
  __shared__ uint32_t part[kSortBlock];
  constexpr uint32_t per = kCells / kSortBlock;
  uint32_t v[per], sum = 0;
  for (uint32_t k = 0; k < per; ++k) { v[k] = hist[threadIdx.x * per + k]; hist[threadIdx.x * per + k] = 0u; sum += v[k]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < kSortBlock; off <<= 1) {
    const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint32_t k = 0; k < per; ++k) { cursor[threadIdx.x * per + k] = run; run += v[k]; }
}

__global__ void __launch_bounds__(kSortBlock) k_cell_scatter(const uint32_t* count_ptr, const uint32_t* q_in, const uint16_t* keys, uint32_t* cursor, uint32_t* q_out)
{
  __shared__ uint32_t h[kCells];     // entries of this block per cell, then the running rank inside the block's range of that cell
  __shared__ uint32_t base[kCells];  // start of this block's range inside the cell's global range
  for (uint32_t b = threadIdx.x; b < kCells; b += kSortBlock) h[b] = 0u;
  __syncthreads();
  uint32_t lo, hi;
  block_share(*count_ptr, lo, hi);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kSortBlock) atomicAdd(&h[keys[i] & (kCells - 1u)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < kCells; b += kSortBlock) {
    base[b] = h[b] ? atomicAdd(&cursor[b], h[b]) : 0u;
    h[b] = 0u;
  }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += kSortBlock) {
    const uint32_t b = keys[i] & (kCells - 1u);
    q_out[base[b] + atomicAdd(&h[b], 1u)] = q_in[i];
  }
}

template <uint32_t LOBES, int BLOCKS = FH_SHADE_BLOCKS>
__global__ void __launch_bounds__(kBlock, (LOBES == L_ALL ? 1 : BLOCKS)) k_shade(SceneDev sc, FrameDev fr, PoolDev pool, uint32_t cls, uint32_t depth)
{
  // (the grid is sized for every path of the pass; the blocks beyond this class's queue leave before they stage anything)
  if (blockIdx.x * blockDim.x >= pool.counters[depth * kCounterStride + CNT_CLS + cls]) return;
  __shared__ SobolRows<4> rows;
  // the shade kernels run one wave per SIMD (512 registers per lane), so nothing hides a dependent global load: the small tables every
  // hit reads -- the two albedo LUTs and, when there are few of them, the material records -- are staged in LDS once per workgroup
  __shared__ float s_lut[kLutReflFloats + kLutSheenFloats];
  __shared__ MaterialDev s_mat[kMatLds];
  for (uint32_t i = threadIdx.x; i < kLutReflFloats; i += blockDim.x) s_lut[i] = fr.lut.reflection[i];
  for (uint32_t i = threadIdx.x; i < kLutSheenFloats; i += blockDim.x) s_lut[kLutReflFloats + i] = fr.lut.sheen[i];
  fr.lut.reflection = s_lut;
  fr.lut.sheen = s_lut + kLutReflFloats;
  if (sc.n_materials <= kMatLds) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.materials);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_mat);
    for (uint32_t i = threadIdx.x; i < sc.n_materials * (uint32_t)(sizeof(MaterialDev) / 4); i += blockDim.x) dst[i] = src[i];
    sc.materials = s_mat;
  }
  __shared__ float s_srgb[256];  // the sRGB decode table: twelve lookups per fetch of a colour texture
  if (sc.n_textures) {
    s_srgb[threadIdx.x] = sc.srgb_lut[threadIdx.x];
    sc.srgb_lut = s_srgb;
  }
  __shared__ HosekSky s_sky;
  stage_sky(fr, s_sky);
  BounceSlots bs;
  bs.set(fr, sc.n_lights, depth);
  bs.load_rows(rows, fr.sobol_bytes);  // (ends with the workgroup barrier that also publishes the tables above)

  uint32_t* cnt = pool.counters + depth * kCounterStride;
  uint32_t* cnt_next = cnt + kCounterStride;
  const uint32_t count = cnt[CNT_CLS + cls];
  const uint32_t* q = pool.q_cls + (size_t)cls * pool.capacity;
  const uint32_t qnext = (depth + 1u) & 1u;
  const uint32_t stride = gridDim.x * blockDim.x;
  __shared__ uint32_t s_reserve[2][10];  // (two areas, used alternately: block_queue_reserve)
  uint32_t iter = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {
    const uint32_t i = base + threadIdx.x;
    const bool valid = i < count;
    bool shaded = false, cont = false;
    uint32_t p = 0, cell = 0;
    if (valid) {
      p = q[i];
      const float4 hit = pool.hit[p];
      PoolSink o(pool, p, hit.x);
      const bool first = depth != 0 || (pool.flags[p] & 4u) == 0u;
      const f3 L = depth == 0u ? mk3(pool.rad[p]) : mk3(0.0f);  // the radiance so far only enters at a directly visible emitter (first hit): later bounces skip the load
      shade_hit<LOBES>(sc, fr, rows, bs, depth, hit, mk3(pool.ray_d[p]), mk3(pool.thr[p]), L, pool.pixel[p], pool.nspp[p], o, first);
      shaded = o.shaded;
      cont = o.cont;
      if (shaded || cont) cell = cell_of(fr, o.origin);
    }
    // queue positions of both appends: one returning atomic per workgroup and queue (fh_device.h: block_queue_reserve)
    uint32_t* const counters2[2] = {&cnt[CNT_SEC], &cnt_next[CNT_RAD]};
    const bool act[2] = {shaded, cont};
    uint32_t pos[2];
    block_queue_reserve<2>(counters2, act, pos, s_reserve[iter & 1u]);
    ++iter;
    if (shaded) { pool.q_sec[pos[0]] = p; pool.key_sec[pos[0]] = (uint16_t)cell; }
    if (cont) { pool.q_rad[qnext][pos[1]] = p; pool.key_rad[pos[1]] = (uint16_t)cell; }
  }
}

// ------------------------------------------------------------------------------------------------
// Finish the BSDF-sampled light ray of a scene with emitters once its closest hit is known
// (pt.cu:952-999 closest-hit light, :531-543 miss light, :910-924 MIS weight).
FH_D f3 resolve_light_ray(const SceneDev& sc, const FrameDev& fr, f3 T, float cosw, f3 f, float pdf, f3 ro, f3 ld, bool hit, const HitRec& h)
{
  f3 le = mk3(0.0f);
  float pdf_light = cosw / kPi;
  bool add = false;
  if (hit) {
    if (sc.face_cls[h.prim] & 0x80u) {
      const size_t fb = kFaceRec * (size_t)h.prim;
      const float4 l0 = sc.face_rec[fb], l1 = sc.face_rec[fb + 1], l2 = sc.face_rec[fb + 2], l3 = sc.face_rec[fb + 3], l4 = sc.face_rec[fb + 4], l5 = sc.face_rec[fb + 5];
      const float lw = 1.0f - h.u - h.v;
      const f3 lp = lw * mk3(l0) + h.u * mk3(l1) + h.v * mk3(l2);
      const f3 ln = lw * mk3(l3) + h.u * mk3(l4) + h.v * mk3(l5);
      if (dot(-ld, ln) > 0.0f) {
        const MaterialDev& lm = sc.materials[__float_as_uint(sc.face_rec[fb + 6].x)];
        le = emission_of(sc, lm, lw * l0.w + h.u * l2.w + h.v * l4.w, lw * l1.w + h.u * l3.w + h.v * l5.w);
        const float area = 0.5f * length(cross(mk3(l1) - mk3(l0), mk3(l2) - mk3(l0)));
        const f3 dl = lp - ro;
        const float r2 = dot(dl, dl);
        const float pdf_area = 1.0f / (sc.n_lights * area);
        pdf_light = r2 / fabsf(dot(-ld, ln)) * pdf_area;
        add = true;
      }
    }
  } else {
    le = env_radiance(fr, ld);
    add = true;
  }
  if (!add) return mk3(0.0f);
  const float w = pdf / (pdf + pdf_light);
  return clamp01(T * w * f * cosw / pdf) * le;
}

// secondary rays, binary-BVH fallback: one thread per shaded path, its rays in the reference's order
template <bool COUNT, bool WIDE, bool LIGHTS, bool ALPHA>
__global__ void __launch_bounds__(kBlock) k_trace_secondary_static(SceneDev sc, FrameDev fr, PoolDev pool, uint32_t depth, TraceCounters tc)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // [entry][thread], sized by the launcher for the depth of the BVH
  __shared__ HosekSky s_sky;  // (light rays that escape see the sky: resolve_light_ray)
  if (LIGHTS) { stage_sky(fr, s_sky); __syncthreads(); }
  const uint32_t count = pool.counters[depth * kCounterStride + CNT_SEC];
  constexpr bool has_lights = LIGHTS;
  uint32_t nn = 0, nt = 0, nr = 0;
  WaveSteps ws;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint32_t p = pool.q_sec[i];
    f3 L = mk3(pool.rad[p]);
#pragma unroll
    for (uint32_t slot = SEC_DIR; slot <= SEC_LIGHT; ++slot) {
      if (slot == SEC_DIR && !fr.has_dir) continue;
      if (slot == SEC_AREA && !has_lights) continue;
      const size_t k = pool.sec_at(slot, p);
      const float4 d = pool.sec[k + 1];
      if (!sec_present(d)) continue;
      const float4 o = pool.sec[k];
      HitRec h;
      if (COUNT) nr++;
      if (slot == SEC_LIGHT && has_lights) {
        const bool hit = WIDE ? traverse_bvh8<false, COUNT, true, ALPHA>(sc.bvh8, mk3(o), mk3(d), o.w, h, nn, nt, &ws, lds_stack, (int)sc.bvh8.depth, &sc)
                              : traverse_bvh2<false, COUNT, ALPHA>(sc.bvh2, mk3(o), mk3(d), o.w, h, nn, nt, &sc);
        const float4 la = pool.lp_a[p], lb = pool.lp_b[p];
        L += resolve_light_ray(sc, fr, mk3(la), la.w, mk3(lb), lb.w, mk3(o), mk3(d), hit, h);
      } else {
        const uint32_t nn0 = nn;
        const bool occluded = WIDE ? traverse_bvh8<true, COUNT, true, ALPHA>(sc.bvh8, mk3(o), mk3(d), o.w, h, nn, nt, &ws, lds_stack, (int)sc.bvh8.depth, &sc)
                                   : traverse_bvh2<true, COUNT, ALPHA>(sc.bvh2, mk3(o), mk3(d), o.w, h, nn, nt, &sc);
        if (COUNT) { const uint32_t kk = nn - nn0; int b = 0; while (b < 7 && kk > (8u << b)) ++b; atomicAdd(tc.hist + b, 1ull); }
        if (!occluded) L += mk3(pool.sec[k + 2]);
      }
    }
    pool.rad[p] = mk4(L, 0.0f);
  }
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)nr);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// secondary rays over the 8-wide BVH with wave-cooperative triangle tests: a lane owns one shaded path and walks its
// secondary-ray slots in the reference's order; a slot is traversed by the whole wave when any lane has a ray in it
template <bool COUNT, bool LIGHTS, bool ALPHA>
__global__ void __launch_bounds__(kBlock, COUNT ? 1 : (LIGHTS ? 5 : 6)) k_trace_secondary_coop(SceneDev sc, FrameDev fr, PoolDev pool, uint32_t depth, TraceCounters tc, uint32_t flush)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // [entry][thread], sized by the launcher for the depth of the BVH
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kBlock / 64) * kCoopLdsBytesPerWave];
  const CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  __shared__ HosekSky s_sky;  // (light rays that escape see the sky: resolve_light_ray)
  if (LIGHTS) { stage_sky(fr, s_sky); __syncthreads(); }
  const uint32_t count = pool.counters[depth * kCounterStride + CNT_SEC];
  constexpr bool has_lights = LIGHTS;
  uint32_t nn = 0, nt = 0, nr = 0;
  WaveSteps ws;
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += gridDim.x * blockDim.x) {
    const uint32_t i = base + (threadIdx.x & 63u);
    const bool in_range = i < count;
    const uint32_t p = in_range ? pool.q_sec[i] : 0u;
    f3 L = in_range ? mk3(pool.rad[p]) : mk3(0.0f);
#pragma unroll
    for (uint32_t slot = SEC_DIR; slot <= SEC_LIGHT; ++slot) {
      if (slot == SEC_DIR && !fr.has_dir) continue;
      if (slot == SEC_AREA && !has_lights) continue;
      const size_t k = pool.sec_at(slot, p);
      const float4 d = in_range ? pool.sec[k + 1] : make_float4(0.0f, 0.0f, 1.0f, 0.0f);
      const bool valid = in_range && sec_present(d);
      if (__ballot(valid) == 0ull) continue;
      const float4 o = valid ? pool.sec[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      HitRec h;
      if (COUNT && valid) nr++;
      if (slot == SEC_LIGHT && has_lights) {
        const bool hit = traverse_bvh8_coop<false, COUNT, true, ALPHA>(sc.bvh8, valid, mk3(o), mk3(d), o.w, h, nn, nt, &ws, cl, flush, lds_stack, (int)sc.bvh8.depth, &sc);
        if (valid) {
          const float4 la = pool.lp_a[p], lb = pool.lp_b[p];
          L += resolve_light_ray(sc, fr, mk3(la), la.w, mk3(lb), lb.w, mk3(o), mk3(d), hit, h);
        }
      } else {
        const uint32_t nn0 = nn;
        const bool occluded = traverse_bvh8_coop<true, COUNT, true, ALPHA>(sc.bvh8, valid, mk3(o), mk3(d), o.w, h, nn, nt, &ws, cl, flush, lds_stack, (int)sc.bvh8.depth, &sc);
        if (COUNT && valid) { const uint32_t kk = nn - nn0; int b = 0; while (b < 7 && kk > (8u << b)) ++b; atomicAdd(tc.hist + b, 1ull); }
        if (valid && !occluded) L += mk3(pool.sec[k + 2]);
      }
    }
    if (in_range) pool.rad[p] = mk4(L, 0.0f);
  }
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)nr);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// secondary rays, streaming: a lane's item is one shaded path; its secondary-ray slots are traced one after the other by the
// same lane, so the additions into the path's radiance keep the reference's order (directional, sky, area, BSDF-sampled)
// (Items that are single RAYS instead of paths -- so that a small launch would end in its longest ray, not in its longest path -- were built and measured in round 6, bit-identical
// and SLOWER: the rays of a path share their first nodes, and the launches of one-pass calls are not bound by a lane's chain.  tools/patches/r6_ray_items.patch, profiles/README.md r6-8.)
template <bool COUNT, bool LIGHTS>
struct SecondaryStream {
  static constexpr bool all_any = !LIGHTS;  // without emitters every secondary ray stops at its first hit
  static constexpr bool can_climb = true;   // first-hit rays may start at the node of the face they leave (fh_trace.h: bottom-up start)
  const SceneDev& sc;
  const FrameDev& fr;
  const PoolDev& pool;
  ChunkFeed feed;
  uint32_t p = 0, slot = 0;
  uint32_t start = 0;  // wide node the lane's current ray starts its traversal at (0: the root)
  bool active = false;
  f3 L;
  uint32_t n_rays = 0;
  unsigned long long* hist;
  HistPack hp;
  FH_D SecondaryStream(const SceneDev& s, const FrameDev& f, const PoolDev& pl, const ChunkFeed& cf, unsigned long long* hs) : sc(s), fr(f), pool(pl), feed(cf), L(mk3(0.0f)), hist(hs) {}
  // first slot >= from of path p that holds a ray
  FH_D bool scan(uint32_t from, f3& o, f3& d, float& tmax, bool& any)
  {
    for (uint32_t s = from; s <= SEC_LIGHT; ++s) {
      if (s == SEC_DIR && !fr.has_dir) continue;
      if (s == SEC_AREA && !LIGHTS) continue;
      const size_t k = pool.sec_at(s, p);
      const float4 d4 = pool.sec[k + 1];
      if (!sec_present(d4)) continue;
      const float4 o4 = pool.sec[k];
      o = mk3(o4); d = mk3(d4); tmax = o4.w;
      start = start_node_of(d4);
      any = !(s == SEC_LIGHT && LIGHTS);
      slot = s;
      if (COUNT) n_rays++;
      return true;
    }
    return false;
  }
  FH_D void finish() { if (active) { pool.rad[p] = mk4(L, 0.0f); active = false; } }
  FH_D bool advance(f3& o, f3& d, float& tmax, bool& any)
  {
    if (!active) return false;
    if (scan(slot + 1u, o, d, tmax, any)) return true;
    finish();
    return false;
  }
  FH_D bool take(uint32_t i, f3& o, f3& d, float& tmax, bool& any)
  {
    if (i >= feed.end) return false;
    return take_item(i, o, d, tmax, any);
  }
  FH_D bool take_item(uint32_t i, f3& o, f3& d, float& tmax, bool& any)
  {
    p = pool.q_sec[i];
    L = mk3(pool.rad[p]);
    active = true;
    if (scan(0u, o, d, tmax, any)) return true;
    active = false;  // a path without secondary rays: its radiance stays as it is
    return false;
  }
  FH_D void commit(bool hit, const HitRec& h, uint32_t nodes)
  {
    const size_t k = pool.sec_at(slot, p);
    if (slot == SEC_LIGHT && LIGHTS) {
      const float4 o = pool.sec[k], d = pool.sec[k + 1];
      const float4 la = pool.lp_a[p], lb = pool.lp_b[p];
      L += resolve_light_ray(sc, fr, mk3(la), la.w, mk3(lb), lb.w, mk3(o), mk3(d), hit, h);
    } else {
      if (COUNT) hp.add(nodes);
      if (!hit) L += mk3(pool.sec[k + 2]);
    }
  }
  FH_D bool drained() const { return feed.drained(); }
  FH_D bool followup() const { return active && slot < SEC_LIGHT; }
  FH_D uint32_t start_node() const { return start; }
};

template <bool COUNT, bool LIGHTS, bool ALPHA>
__global__ void __launch_bounds__(kBlock, COUNT ? 1 : (LIGHTS ? FH_SECONDARY_BLOCKS_HEAVY : (ALPHA ? FH_STREAM_BLOCKS_ALPHA : FH_STREAM_BLOCKS))) k_trace_secondary_stream(SceneDev sc, FrameDev fr, PoolDev pool, uint32_t depth, TraceCounters tc, uint32_t flush, uint32_t refill, uint32_t chunk, uint32_t min_rays, StackSpill spill)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // [entry][thread], sized by the launcher for the depth of the BVH
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kBlock / 64) * kCoopLdsBytesPerWave];
  const uint32_t count = pool.counters[depth * kCounterStride + CNT_SEC];
  if (stream_block_idle(count, min_rays)) return;
  __shared__ HosekSky s_sky;  // (light rays that escape see the sky: resolve_light_ray)
  if (LIGHTS) { stage_sky(fr, s_sky); __syncthreads(); }
  const ClockStamp stamp;
  CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  AlphaLds<AlphaDefer<true, ALPHA>::value>::attach(cl);  // candidates waiting for their any-hit test (fh_trace.h: alpha_ring): LDS of the kernels with the test compiled in only
  uint32_t nn = 0, nt = 0;
  WaveSteps ws;
  SecondaryStream<COUNT, LIGHTS> pol(sc, fr, pool, ChunkFeed(pool.counters + depth * kCounterStride + CNT_CUR_SEC, count, stream_chunk_for(count, chunk)), tc.hist);
  __shared__ uint4 lds_top[kTopNodes * 4];
  stage_top_nodes(sc.bvh8, lds_top);
  traverse_stream<true, COUNT, true, ALPHA>(sc.bvh8, pol, nn, nt, &ws, cl, flush, refill, lds_stack, (int)sc.bvh8.depth, &sc, spill, lds_top,
                                            spill.probe ? pool.counters + depth * kCounterStride + CNT_COST_NODE : nullptr);
  pol.finish();
  stamp.commit(tc.clk);
  if (COUNT) {
    atomicAdd(tc.nodes, (unsigned long long)nn);
    atomicAdd(tc.tris, (unsigned long long)nt);
    atomicAdd(tc.rays, (unsigned long long)pol.n_rays);
    pol.hp.flush(tc.hist);
    if (ws.node) atomicAdd(tc.wave_nodes, (unsigned long long)ws.node);
    if (ws.tri) atomicAdd(tc.wave_tris, (unsigned long long)ws.tri);
  }
}

// One launch for the secondary rays of bounce b AND the closest-hit rays of bounce b + 1 (calls of ONE pass: the reference's 1- and 16-sample calls).  The two do not
// depend on each other -- the secondary rays read the secondary-ray records and add to the radiance, the closest-hit rays read the path's ray and write its hit -- and as two
// launches each of them ends in its own few long rays with most of the chip idle.  Here the wave's work queue is the secondary queue followed by the next bounce's radiance
// queue (one cursor over both): a lane holds either a path with its secondary rays or one closest-hit ray, every ray stops at its first hit or not as its kind says
// (traverse_stream's per-lane flag), and the launch has ONE end.  Same device functions, same per-path order of operations: the bits do not change.
template <bool LIGHTS>
struct MergedStream {
  static constexpr bool all_any = false;
  static constexpr bool can_climb = false;
  SecondaryStream<false, LIGHTS> sec;
  const PoolDev& next;        // the view whose radiance queue of bounce b + 1 is traced
  const uint32_t* q_closest;
  uint32_t n_sec;
  ChunkFeed feed;
  uint32_t pc = 0, qc = 0;    // path slot and entry (in the next bounce's radiance queue) of the lane's closest-hit ray
  uint32_t start_closest = 0;
  bool closest = false;       // the lane's current item is a closest-hit ray
  FH_D MergedStream(const SceneDev& s, const FrameDev& f, const PoolDev& ps, const PoolDev& pn, const uint32_t* qc, uint32_t ns, const ChunkFeed& cf)
      : sec(s, f, ps, cf, nullptr), next(pn), q_closest(qc), n_sec(ns), feed(cf) {}
  FH_D bool advance(f3& o, f3& d, float& tmax, bool& any) { return closest ? false : sec.advance(o, d, tmax, any); }
  FH_D bool take(uint32_t i, f3& o, f3& d, float& tmax, bool& any)
  {
    if (i >= feed.end) return false;
    if (i < n_sec) { closest = false; return sec.take_item(i, o, d, tmax, any); }
    closest = true;
    qc = i - n_sec;
    pc = q_closest[qc];
    const float4 o4 = next.ray_o[pc], d4 = next.ray_d[pc];
    o = mk3(o4); d = mk3(d4); tmax = o4.w; any = false;
    start_closest = start_node_of(d4);
    return true;
  }
  FH_D uint32_t start_node() const { return closest ? start_closest : sec.start; }
  FH_D void commit(bool hit, const HitRec& h, uint32_t nodes)
  {
    if (closest) { next.hit[pc] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim)); next.q_prim[qc] = h.prim; }
    else sec.commit(hit, h, nodes);
  }
  FH_D bool drained() const { return feed.drained(); }
  FH_D bool followup() const { return !closest && sec.followup(); }
};

template <bool LIGHTS, bool ALPHA>
__global__ void __launch_bounds__(kBlock, LIGHTS ? FH_SECONDARY_BLOCKS_HEAVY : (ALPHA ? FH_STREAM_BLOCKS_ALPHA : FH_STREAM_BLOCKS)) k_trace_merged_stream(SceneDev sc, FrameDev fr, PoolDev ps, PoolDev pn, uint32_t depth, TraceCounters tc, uint32_t flush, uint32_t refill, uint32_t chunk, uint32_t min_rays, StackSpill spill)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
  __shared__ __attribute__((aligned(16))) unsigned char lds[(kBlock / 64) * kCoopLdsBytesPerWave];
  const uint32_t n_sec = ps.counters[depth * kCounterStride + CNT_SEC], n_closest = ps.counters[(depth + 1u) * kCounterStride + CNT_RAD];
  const uint32_t count = n_sec + n_closest;
  if (stream_block_idle(count, min_rays)) return;
  __shared__ HosekSky s_sky;
  if (LIGHTS) { stage_sky(fr, s_sky); __syncthreads(); }
  const ClockStamp stamp;
  CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  AlphaLds<AlphaDefer<true, ALPHA>::value>::attach(cl);  // candidates waiting for their any-hit test (fh_trace.h: alpha_ring): LDS of the kernels with the test compiled in only
  uint32_t nn = 0, nt = 0;
  MergedStream<LIGHTS> pol(sc, fr, ps, pn, pn.q_rad[(depth + 1u) & 1u], n_sec, ChunkFeed(ps.counters + depth * kCounterStride + CNT_CUR_SEC, count, stream_chunk_for(count, chunk)));
  __shared__ uint4 lds_top[kTopNodes * 4];
  stage_top_nodes(sc.bvh8, lds_top);
  traverse_stream<true, false, true, ALPHA>(sc.bvh8, pol, nn, nt, nullptr, cl, flush, refill, lds_stack, (int)sc.bvh8.depth, &sc, spill, lds_top,
                                            spill.probe ? ps.counters + depth * kCounterStride + CNT_COST_NODE : nullptr);
  pol.sec.finish();
  stamp.commit(tc.clk);
}

// ------------------------------------------------------------------------------------------------
// Tail: after the first bounces only a few thousand of the millions of paths of a pass are still alive,
// and a bounce-synchronous wavefront then costs one worst-case ray latency per kernel and bounce.
// k_tail finishes those paths in one launch: each lane carries its path through all remaining bounces
// (trace -> shade -> secondary rays -> Russian roulette), so slow rays of different paths overlap
// instead of adding up.  Same device functions, same per-path operation order as the wavefront
// kernels, hence the same bits.
// One instantiation per lobe set, like k_shade: the generic seven-lobe form needs 430 registers (one wave per SIMD, so the 128 Ki paths a small pass hands over took two
// rounds), the forms for the lobe sets real scenes have fit two waves per SIMD.
//
// The rays of a bounce are traced PACKED.  A path in the tail has up to five rays per bounce -- its secondary rays and, once it is shaded, the closest-hit ray of the
// NEXT bounce, which depends on none of them -- and by its third bounce most lanes of a wave have no path left.  Traced one kind after the other, each kind costs the wave its
// longest ray (~1 us per dependent step with the SIMD to itself), three to five times per bounce; instead all rays of the wave are numbered (kind by kind, lane by lane:
// ballot + popcount), staged in LDS 64 at a time, traced by whichever lanes the numbering gives them to -- each ray stops at its first hit or not as its kind says -- and the
// results go back to their owners through LDS.  The additions into a path's radiance stay in the reference's order (they are applied from the returned results, slot by slot), so
// the bits do not change; the chain of a bounce is one traversal and one shade instead of three traversals and one shade once a wave is sparse.
constexpr uint32_t kTailRays = SEC_COUNT + 1u;  // the secondary-ray slots and the next bounce's closest-hit ray
static_assert(kTailRays == 5u, "trace_all spells its per-lane fallback out ray by ray");
__device__ __attribute__((noinline)) bool tail_trace_lane(const SceneDev& sc, uint2* lds_stack, bool any, f3 o, f3 d, float tmax, HitRec& h)
{
  uint32_t a = 0, b = 0;
  if (!sc.use_bvh8) return any ? traverse<true, false>(sc, o, d, tmax, h, a, b) : traverse<false, false>(sc, o, d, tmax, h, a, b);
  if (sc.has_alpha) return any ? traverse_bvh8<true, false, true, true>(sc.bvh8, o, d, tmax, h, a, b, nullptr, lds_stack, (int)sc.bvh8.depth, &sc)
                               : traverse_bvh8<false, false, true, true>(sc.bvh8, o, d, tmax, h, a, b, nullptr, lds_stack, (int)sc.bvh8.depth, &sc);
  return any ? traverse_bvh8<true, false, true, false>(sc.bvh8, o, d, tmax, h, a, b, nullptr, lds_stack, (int)sc.bvh8.depth, &sc)
             : traverse_bvh8<false, false, true, false>(sc.bvh8, o, d, tmax, h, a, b, nullptr, lds_stack, (int)sc.bvh8.depth, &sc);
}
struct TailRay { bool has = false; bool any = true; f3 o = mk3(0.0f), d = mk3(0.0f, 0.0f, 1.0f); float tmax = 0.0f; HitRec h = HitRec{0.0f, 0.0f, 0.0f, 0xffffffffu}; bool hit = false; };  // (no indeterminate field: r6-13)

template <uint32_t LOBES>
__global__ void __launch_bounds__(kBlock, LOBES == L_ALL ? 1 : 2) k_tail(SceneDev sc, FrameDev fr, PoolDev pool, uint32_t first_depth, uint32_t coop_flush)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // traversal stack of every lane ([entry][thread], as in the streaming kernels): a private array lands in scratch
  __shared__ __attribute__((aligned(16))) unsigned char lds_coop[(kBlock / 64) * kCoopLdsBytesPerWave];
  __shared__ float4 s_in[kBlock / 64][2][64];  // one round of packed rays of a wave: (origin, tmax), (direction, stops at its first hit)
  __shared__ float4 s_out[kBlock / 64][64];    // and their hits: t, u, v, face id bits
  if (blockIdx.x * blockDim.x >= pool.counters[first_depth * kCounterStride + CNT_RAD]) return;  // (the grid is sized for the pass, the survivors are few: most blocks have nothing to do, and leave before the tables are staged)
  const CoopLds cl = coop_lds(lds_coop, threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const bool coop = sc.use_bvh8 && coop_flush != 0u;
  // all rays of the wave (called by its 64 lanes together)
  auto trace_all = [&](TailRay (&rays)[kTailRays]) {
    if (!coop) {  // one ray per lane, per-lane loops: tiny scenes (binary tree) and FH_COOP=0 (an out-of-line call per ray: five inlined copies of every traversal would be most of the kernel)
      HitRec t;  // (only this one has its address taken: the rays themselves stay in registers)
      if (rays[0].has) { rays[0].hit = tail_trace_lane(sc, lds_stack, rays[0].any, rays[0].o, rays[0].d, rays[0].tmax, t); rays[0].h = t; }
      if (rays[1].has) { rays[1].hit = tail_trace_lane(sc, lds_stack, rays[1].any, rays[1].o, rays[1].d, rays[1].tmax, t); rays[1].h = t; }
      if (rays[2].has) { rays[2].hit = tail_trace_lane(sc, lds_stack, rays[2].any, rays[2].o, rays[2].d, rays[2].tmax, t); rays[2].h = t; }
      if (rays[3].has) { rays[3].hit = tail_trace_lane(sc, lds_stack, rays[3].any, rays[3].o, rays[3].d, rays[3].tmax, t); rays[3].h = t; }
      if (rays[4].has) { rays[4].hit = tail_trace_lane(sc, lds_stack, rays[4].any, rays[4].o, rays[4].d, rays[4].tmax, t); rays[4].h = t; }
      return;
    }
    uint32_t idx[kTailRays], total = 0;  // position of this lane's ray of kind k among the wave's rays; total: wave-uniform
#pragma unroll
    for (uint32_t k = 0; k < kTailRays; ++k) {
      const unsigned long long m = __ballot(rays[k].has);
      idx[k] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      total += (uint32_t)__popcll(m);
    }
    for (uint32_t lo = 0; lo < total; lo += 64u) {
#pragma unroll
      for (uint32_t k = 0; k < kTailRays; ++k)
        if (rays[k].has && idx[k] - lo < 64u) {
          s_in[wave][0][idx[k] - lo] = mk4(rays[k].o, rays[k].tmax);
          s_in[wave][1][idx[k] - lo] = mk4(rays[k].d, rays[k].any ? 1.0f : 0.0f);
        }
      const bool valid = lo + lane < total;
      const float4 a = valid ? s_in[wave][0][lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = valid ? s_in[wave][1][lane] : make_float4(0.0f, 0.0f, 1.0f, 0.0f);
      HitRec h;
      uint32_t na = 0, nb = 0;
      if (sc.has_alpha) traverse_bvh8_coop_mode<2, false, true, true>(sc.bvh8, valid, b.w != 0.0f, mk3(a), mk3(b), a.w, h, na, nb, nullptr, cl, coop_flush, lds_stack, (int)sc.bvh8.depth, &sc);
      else traverse_bvh8_coop_mode<2, false, true, false>(sc.bvh8, valid, b.w != 0.0f, mk3(a), mk3(b), a.w, h, na, nb, nullptr, cl, coop_flush, lds_stack, (int)sc.bvh8.depth, &sc);
      if (valid) s_out[wave][lane] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
#pragma unroll
      for (uint32_t k = 0; k < kTailRays; ++k)
        if (rays[k].has && idx[k] - lo < 64u) {
          const float4 r = s_out[wave][idx[k] - lo];
          rays[k].h.t = r.x; rays[k].h.u = r.y; rays[k].h.v = r.z; rays[k].h.prim = __float_as_uint(r.w);
          rays[k].hit = rays[k].h.prim != 0xffffffffu;
        }
    }
  };
  __shared__ SobolRows<4> rows;
  __shared__ float s_lut[kLutReflFloats + kLutSheenFloats];  // as in k_shade: small tables every hit reads live in LDS
  __shared__ MaterialDev s_mat[kMatLds];
  for (uint32_t i = threadIdx.x; i < kLutReflFloats; i += blockDim.x) s_lut[i] = fr.lut.reflection[i];
  for (uint32_t i = threadIdx.x; i < kLutSheenFloats; i += blockDim.x) s_lut[kLutReflFloats + i] = fr.lut.sheen[i];
  fr.lut.reflection = s_lut;
  fr.lut.sheen = s_lut + kLutReflFloats;
  if (sc.n_materials <= kMatLds) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.materials);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_mat);
    for (uint32_t i = threadIdx.x; i < sc.n_materials * (uint32_t)(sizeof(MaterialDev) / 4); i += blockDim.x) dst[i] = src[i];
    sc.materials = s_mat;
  }
  __shared__ HosekSky s_sky;
  stage_sky(fr, s_sky);
  __syncthreads();
  const uint32_t count = pool.counters[first_depth * kCounterStride + CNT_RAD];
  const uint32_t* q = pool.q_rad[first_depth & 1u];
  const bool has_lights = sc.n_lights > 0;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {  // (block-uniform bound: the bounce loop holds workgroup barriers)
    const uint32_t i = base + threadIdx.x;
    bool alive = i < count;
    uint32_t p = 0, image_idx = 0, n_spp = 0;
    f3 ro = mk3(0.0f), rd = mk3(0.0f, 0.0f, 1.0f), T = mk3(0.0f), L = mk3(0.0f);
    if (alive) {
      p = q[i];
      const float4 o = pool.ray_o[p];
      ro = mk3(o); rd = mk3(pool.ray_d[p]); T = mk3(pool.thr[p]); L = mk3(pool.rad[p]);
      image_idx = pool.pixel[p]; n_spp = pool.nspp[p];
    }
    // the closest hit of the first bounce; every later one comes back with the rays of the bounce before it
    HitRec h;
    bool hit = false;
    {
      TailRay rays[kTailRays];
      rays[SEC_COUNT].has = alive; rays[SEC_COUNT].any = false; rays[SEC_COUNT].o = ro; rays[SEC_COUNT].d = rd; rays[SEC_COUNT].tmax = 1e9f;
      trace_all(rays);
      h = rays[SEC_COUNT].h; hit = rays[SEC_COUNT].hit;
    }
    for (uint32_t depth = first_depth; depth < fr.max_depth; ++depth) {
      BounceSlots bs;
      bs.set(fr, sc.n_lights, depth);
      __syncthreads();  // rows of the previous bounce are no longer read
      bs.load_rows(rows, fr.sobol_bytes);
      if (__ballot(alive) == 0ull) continue;  // (wave-uniform; the barriers above are still met)
      ShadeOut o;
      if (alive) {
        if (!hit) alive = false;  // pt.cu:504-523 with firsthit == false: nothing added
        else shade_hit<LOBES>(sc, fr, rows, bs, depth, make_float4(h.t, h.u, h.v, __uint_as_float(h.prim)), rd, T, L, image_idx, n_spp, o);
      }
      TailRay rays[kTailRays];
#pragma unroll
      for (uint32_t slot = SEC_DIR; slot <= SEC_LIGHT; ++slot) {
        if (slot == SEC_DIR && !fr.has_dir) continue;
        if (slot == SEC_AREA && !has_lights) continue;
        rays[slot].has = alive && o.sec[slot].active;
        rays[slot].any = !(slot == SEC_LIGHT && has_lights);
        rays[slot].o = o.sec[slot].o; rays[slot].d = o.sec[slot].d; rays[slot].tmax = o.sec[slot].tmax;
      }
      rays[SEC_COUNT].has = alive && o.cont; rays[SEC_COUNT].any = false; rays[SEC_COUNT].o = o.next_o; rays[SEC_COUNT].d = o.next_d; rays[SEC_COUNT].tmax = 1e9f;
      trace_all(rays);
      // secondary rays in the reference's order
#pragma unroll
      for (uint32_t slot = SEC_DIR; slot <= SEC_LIGHT; ++slot) {
        if (!rays[slot].has) continue;
        if (slot == SEC_LIGHT && has_lights) L += resolve_light_ray(sc, fr, o.lp_T, o.lp_cos, o.lp_f, o.lp_pdf, o.sec[slot].o, o.sec[slot].d, rays[slot].hit, rays[slot].h);
        else if (!rays[slot].hit) L += o.sec[slot].c;
      }
      if (alive) {
        if (o.cont) { rd = o.next_d; T = o.T; h = rays[SEC_COUNT].h; hit = rays[SEC_COUNT].hit; }
        else alive = false;
      }
    }
    if (i < count) pool.rad[p] = mk4(L, 0.0f);
  }
}

// ------------------------------------------------------------------------------------------------
// FH_FLAG_REFERENCE_FIRSTHIT: the reference declares its payload outside the per-launch sample loop and never resets `firsthit`
// (pt.cu:432-433), so within ONE launch of n_samples > 1 only the first sample that hits anything records AOVs and sees emitters directly
// (:745-760), later primary misses add no sky (:509), and every sample averages the AOVs of that first hit again (:483-487).  The state is
// one bit per pixel, carried across the passes of a launch: after the depth-0 trace this kernel walks each pixel's samples in order and marks
// the paths that come after the first hitting one (flag 4).
__global__ void __launch_bounds__(kBlock) k_firsthit_scan(PoolDev pool, const uint32_t* owned, uint32_t n_owned, uint32_t n_batch, uint32_t* seen_state)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_owned; i += gridDim.x * blockDim.x) {
    const uint32_t image_idx = owned[i];
    bool seen = seen_state[image_idx] != 0u;
    for (uint32_t k = 0; k < n_batch; ++k) {
      const uint32_t p = k * n_owned + i;
      const uint32_t fl = pool.flags[p];
      const bool in_queue = (fl & 2u) == 0u;  // otherwise the path ended in k_generate (missed the scene bounds): a primary miss
      if (seen) {
        pool.flags[p] = fl | 4u;
        if (!in_queue) pool.rad[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // its sky contribution is dropped (pt.cu:509)
      } else if (in_queue && __float_as_uint(pool.hit[p].w) != 0xffffffffu) {
        seen = true;  // this sample is the launch's first hit: it behaves normally
      }
    }
    seen_state[image_idx] = seen ? 1u : 0u;
  }
}

// ------------------------------------------------------------------------------------------------
template <bool QUIRK>
__global__ void __launch_bounds__(kBlock) k_accumulate(PoolDev pool, LayersDev layers, const uint32_t* owned, uint32_t n_owned, uint32_t n_batch, float4* carry)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_owned; i += gridDim.x * blockDim.x) {
    const uint32_t image_idx = owned[i];
    uint32_t n_spp = layers.sample_count[image_idx];
    f3 beauty = mk3(layers.beauty[image_idx]), position = mk3(layers.position[image_idx]), normal = mk3(layers.normal[image_idx]), albedo = mk3(layers.albedo[image_idx]);
    float depth = layers.depth[image_idx];
    const float4 tc4 = layers.texcoord[image_idx];
    float tcx = tc4.x, tcy = tc4.y;
    // bug-compat mode: the payload's AOVs persist from the launch's first hit (pt.cu:483-487); carried across the passes of the launch
    f3 apos = mk3(0.0f), anrm = mk3(0.0f), aalb = mk3(0.0f);
    float au = 0.0f, av = 0.0f, ad = 0.0f;
    if (QUIRK) {
      apos = mk3(carry[4 * (size_t)image_idx]); anrm = mk3(carry[4 * (size_t)image_idx + 1]); aalb = mk3(carry[4 * (size_t)image_idx + 2]);
      const float4 td = carry[4 * (size_t)image_idx + 3];
      au = td.x; av = td.y; ad = td.z;
    }
    for (uint32_t k = 0; k < n_batch; ++k) {
      const uint32_t p = k * n_owned + i;
      const f3 L = mk3(pool.rad[p]);
      const f3 radiance = bad3(L) ? mk3(0.0f) : L;
      if (!QUIRK) { apos = mk3(0.0f); anrm = mk3(0.0f); aalb = mk3(0.0f); au = av = ad = 0.0f; }
      if (pool.flags[p] & 1u) {
        apos = mk3(pool.aov_position[p]);
        anrm = mk3(pool.aov_normal[p]);
        aalb = mk3(pool.aov_albedo[p]);
        const float4 td = pool.aov_texdepth[p];
        au = td.x; av = td.y; ad = td.z;
      }
      const float coef = 1.0f / (n_spp + 1.0f);
      const float fn = (float)n_spp;
      beauty = coef * (fn * beauty + radiance);
      position = coef * (fn * position + apos);
      normal = coef * (fn * normal + anrm);
      depth = coef * (fn * depth + ad);
      tcx = coef * (fn * tcx + au);
      tcy = coef * (fn * tcy + av);
      albedo = coef * (fn * albedo + aalb);
      n_spp++;
    }
    if (QUIRK) {
      carry[4 * (size_t)image_idx] = mk4(apos, 0.0f); carry[4 * (size_t)image_idx + 1] = mk4(anrm, 0.0f); carry[4 * (size_t)image_idx + 2] = mk4(aalb, 0.0f);
      carry[4 * (size_t)image_idx + 3] = make_float4(au, av, ad, 0.0f);
    }
    layers.sample_count[image_idx] = n_spp;
    layers.beauty[image_idx] = mk4(beauty, 1.0f);
    layers.position[image_idx] = mk4(position, 1.0f);
    layers.normal[image_idx] = mk4(normal, 1.0f);
    layers.depth[image_idx] = depth;
    layers.texcoord[image_idx] = make_float4(tcx, tcy, 0.0f, 1.0f);
    layers.albedo[image_idx] = mk4(albedo, 1.0f);
  }
}

// counting sort of one bounce queue by the cell keys stored next to it (kernels above); `bins` = histogram (zero on entry and exit) + cursors
void sort_queue_by_cell(hipStream_t st, uint32_t blocks, const uint32_t* count_ptr, const uint32_t* q_in, const uint16_t* keys, uint32_t* bins, uint32_t* q_out)
{
  hipLaunchKernelGGL(k_cell_hist, dim3(blocks), dim3(kSortBlock), 0, st, count_ptr, keys, bins);
  hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(kSortBlock), 0, st, bins, bins + kCells);
  hipLaunchKernelGGL(k_cell_scatter, dim3(blocks), dim3(kSortBlock), 0, st, count_ptr, q_in, keys, bins + kCells, q_out);
}

template <typename F>
void with_bool(bool b, F&& f)
{
  if (b) f(std::true_type{});
  else f(std::false_type{});
}

uint32_t grid_for(uint32_t n)  // multiple of 8 (one share per XCD), at most 8192 blocks, grid-stride beyond
{
  uint32_t b = (n + kBlock - 1) / kBlock;
  b = (b + 7u) & ~7u;
  return b < 8 ? 8 : (b > 8192 ? 8192 : b);
}

template <uint32_t LOBES>
void launch_shade(hipStream_t st, uint32_t grid, const SceneDev& sc, const FrameDev& fr, const PoolDev& pool, uint32_t cls, uint32_t depth, bool three)
{
  if (three && LOBES != L_ALL) hipLaunchKernelGGL((k_shade<LOBES, (LOBES == L_ALL ? FH_SHADE_BLOCKS : 3)>), dim3(grid), dim3(kBlock), 0, st, sc, fr, pool, cls, depth);
  else hipLaunchKernelGGL(k_shade<LOBES>, dim3(grid), dim3(kBlock), 0, st, sc, fr, pool, cls, depth);
}

void dispatch_shade(hipStream_t st, uint32_t grid, uint32_t lobes, const SceneDev& sc, const FrameDev& fr, const PoolDev& pool, uint32_t cls, uint32_t depth, bool three)
{
  // compiled variants, most specific first; a variant is usable when it contains every lobe the class needs
  if ((lobes & ~(uint32_t)L_DIFF) == 0) return launch_shade<L_DIFF>(st, grid, sc, fr, pool, cls, depth, three);
  if ((lobes & ~(uint32_t)L_METAL) == 0) return launch_shade<L_METAL>(st, grid, sc, fr, pool, cls, depth, three);
  if ((lobes & ~(uint32_t)(L_SPEC | L_DIFF)) == 0) return launch_shade<L_SPEC | L_DIFF>(st, grid, sc, fr, pool, cls, depth, three);
  if ((lobes & ~(uint32_t)(L_METAL | L_SPEC | L_DIFF)) == 0) return launch_shade<L_METAL | L_SPEC | L_DIFF>(st, grid, sc, fr, pool, cls, depth, three);  // glTF metallic-roughness materials
  if ((lobes & ~(uint32_t)(L_COAT | L_METAL | L_SPEC | L_DIFF)) == 0) return launch_shade<L_COAT | L_METAL | L_SPEC | L_DIFF>(st, grid, sc, fr, pool, cls, depth, three);  // ... with KHR_materials_clearcoat
  return launch_shade<L_ALL>(st, grid, sc, fr, pool, cls, depth, three);
}

// registers / LDS / scratch / resident workgroups per CU of the shade kernel a class of `lobes` is shaded by (fh_kernel_info, which >= 2): the same choice as dispatch_shade
template <uint32_t LOBES>
hipError_t shade_attributes_of(bool three, hipFuncAttributes& at, int& blocks)
{
  const void* fn = (three && LOBES != L_ALL) ? (const void*)k_shade<LOBES, (LOBES == L_ALL ? FH_SHADE_BLOCKS : 3)> : (const void*)k_shade<LOBES>;
  const hipError_t e = hipFuncGetAttributes(&at, fn);
  if (e != hipSuccess) return e;
  if (three && LOBES != L_ALL) return hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_shade<LOBES, (LOBES == L_ALL ? FH_SHADE_BLOCKS : 3)>, kBlock, 0);
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_shade<LOBES>, kBlock, 0);
}
hipError_t shade_attributes(uint32_t lobes, bool three, hipFuncAttributes& at, int& blocks, uint32_t& compiled_lobes)
{
  if ((lobes & ~(uint32_t)L_DIFF) == 0) { compiled_lobes = L_DIFF; return shade_attributes_of<L_DIFF>(three, at, blocks); }
  if ((lobes & ~(uint32_t)L_METAL) == 0) { compiled_lobes = L_METAL; return shade_attributes_of<L_METAL>(three, at, blocks); }
  if ((lobes & ~(uint32_t)(L_SPEC | L_DIFF)) == 0) { compiled_lobes = L_SPEC | L_DIFF; return shade_attributes_of<L_SPEC | L_DIFF>(three, at, blocks); }
  if ((lobes & ~(uint32_t)(L_METAL | L_SPEC | L_DIFF)) == 0) { compiled_lobes = L_METAL | L_SPEC | L_DIFF; return shade_attributes_of<L_METAL | L_SPEC | L_DIFF>(three, at, blocks); }
  if ((lobes & ~(uint32_t)(L_COAT | L_METAL | L_SPEC | L_DIFF)) == 0) { compiled_lobes = L_COAT | L_METAL | L_SPEC | L_DIFF; return shade_attributes_of<L_COAT | L_METAL | L_SPEC | L_DIFF>(three, at, blocks); }
  compiled_lobes = L_ALL;
  return shade_attributes_of<L_ALL>(three, at, blocks);
}

hipEvent_t take_event(fh_ctx* ctx)
{
  if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct Span {
  fh_ctx* ctx; hipStream_t st; int kind; hipEvent_t a = nullptr, b = nullptr; bool on;
  Span(fh_ctx* c, hipStream_t s, int k) : ctx(c), st(s), kind(k), on((c->flags & FH_FLAG_TIME_KERNELS) != 0)
  {
    if (on) { a = take_event(ctx); b = take_event(ctx); (void)hipEventRecord(a, st); }
  }
  ~Span()
  {
    if (on) { (void)hipEventRecord(b, st); ctx->spans.push_back({a, b, kind}); }
  }
};

}  // namespace

SceneDev scene_dev(const fh_ctx* ctx)
{
  SceneDev s{};
  s.face_rec = ctx->d_face_rec;
  s.face_cls = ctx->d_face_cls;
  s.materials = ctx->d_materials;
  s.lights = ctx->d_lights;
  s.n_faces = ctx->n_faces;
  s.n_lights = ctx->n_lights;
  s.n_materials = ctx->n_materials;
  s.textures = ctx->d_textures;
  s.srgb_lut = ctx->d_srgb_lut;
  s.n_textures = ctx->n_textures;
  s.alpha_rec = ctx->d_alpha_rec;
  s.has_alpha = (ctx->has_alpha || ctx->tun.force_alpha) ? 1u : 0u;
  s.bvh2.nodes = ctx->d_bvh2_nodes;
  s.bvh2.tris = ctx->d_bvh2_tris;
  s.bvh2.n_nodes = ctx->bvh2_n_nodes;
  s.bvh2.n_tris = ctx->bvh2_n_tris;
  s.bvh8.nodes = ctx->d_bvh8_nodes;
  s.bvh8.tris = ctx->d_bvh8_tris;
  s.bvh8.n_nodes = ctx->bvh8_n_nodes;
  s.bvh8.n_tris = ctx->bvh8_n_tris;
  s.bvh8.depth = stack_entries_for(ctx->bvh8_depth);
  s.use_bvh8 = ctx->use_bvh8 ? 1u : 0u;
  // bottom-up start (fh_trace.h): rays that leave a surface begin at the wide node that holds the face.  Only the streaming kernels climb; they trace trees of 4096 nodes and more
  // (what the scene CAN do; render_submit switches it per pass: forced by FH_BOTTOM_UP, else by what the first passes of the scene measure)
  const bool bottom_up = ctx->tun.bottom_up != 0 && ctx->use_bvh8 && ctx->d_bvh8_parent && ctx->d_face_node && ctx->tun.coop && ctx->tun.stream && (ctx->tun.stream_forced || ctx->bvh8_n_nodes >= 4096u) &&
                         ctx->bvh8_n_tris < kCoopMaxTris && !((ctx->has_alpha || ctx->tun.force_alpha));
  s.bvh8.parent = bottom_up ? ctx->d_bvh8_parent : nullptr;
  s.face_node = bottom_up ? ctx->d_face_node : nullptr;
  return s;
}

// fh_kernel_info: registers / static LDS / scratch of the streaming kernel variant this scene is traced by, and how it is launched
int kernel_info(fh_ctx* ctx, int which, uint32_t out[6])
{
  for (int k = 0; k < 6; ++k) out[k] = 0u;
  if (which >= 2) {  // the shade kernel of shading class which - 2 of the scene: out[3] = resident workgroups per CU (x 4 waves / 4 SIMDs = waves per SIMD), out[4] = its lobe mask as compiled
    const uint32_t c = (uint32_t)(which - 2);
    if (c >= ctx->n_classes) return fail(ctx, FH_E_INVALID, "fh_kernel_info: the scene has no such shading class");
    hipFuncAttributes sa{};
    int blocks = 0;
    uint32_t compiled = 0;
    const bool three = ctx->tun.shade_wgs ? ctx->tun.shade_wgs == 3u : true;
    const hipError_t se = shade_attributes(ctx->class_lobes[c], three, sa, blocks, compiled);
    if (se != hipSuccess) return fail(ctx, FH_E_HIP, std::string("hipFuncGetAttributes: ") + hipGetErrorString(se));
    out[0] = (uint32_t)sa.numRegs; out[1] = (uint32_t)sa.sharedSizeBytes; out[2] = (uint32_t)sa.localSizeBytes; out[3] = (uint32_t)(blocks > 0 ? blocks : 0); out[4] = compiled; out[5] = ctx->class_lobes[c];
    return FH_OK;
  }
  const bool alpha = ctx->has_alpha || ctx->tun.force_alpha;
  hipFuncAttributes at{};
  hipError_t e = hipSuccess;
  with_bool(alpha, [&](auto A) {
    if (which == 0) e = hipFuncGetAttributes(&at, (const void*)k_trace_closest_stream<false, decltype(A)::value>);
    else with_bool(ctx->n_lights > 0, [&](auto Li) { e = hipFuncGetAttributes(&at, (const void*)k_trace_secondary_stream<false, decltype(Li)::value, decltype(A)::value>); });
  });
  if (e != hipSuccess) return fail(ctx, FH_E_HIP, std::string("hipFuncGetAttributes: ") + hipGetErrorString(e));
  out[0] = (uint32_t)at.numRegs;
  out[1] = (uint32_t)at.sharedSizeBytes;
  out[2] = (uint32_t)at.localSizeBytes;
  out[3] = ctx->info_blocks[which];
  out[4] = ctx->info_entries[which];
  out[5] = stack_entries_for(ctx->bvh8_depth);
  return FH_OK;
}

void pool_release(fh_ctx* ctx)
{
  (void)hipStreamSynchronize(ctx->stream);  // nothing may still be running out of the buffers (the counter snapshots trail the accumulate)
  for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(ctx->aux_stream[k]);
  if (ctx->sky_stream) (void)hipStreamSynchronize(ctx->sky_stream);
  for (int k = 0; k < 3; ++k) {
    for (void* p : ctx->pool_allocs[k]) (void)hipFree(p);
    ctx->pool_allocs[k].clear();
    ctx->pool_alloc_bytes[k] = 0;
    ctx->pool[k] = PoolDev{};
    ctx->pool_shape[k] = fh_ctx::PoolShape{};
    ctx->counters_in_flight[k] = false;
  }
}

int pool_ensure(fh_ctx* ctx, int slot, uint32_t capacity)
{
  // what a path record has to hold depends on the scene and the lights: a place per kind of secondary ray, the pending light-ray record (emitters
  // only), one queue per shading class.  A pool only grows: in paths, or when a later frame needs a kind of ray / a class it has no place for.
  fh_ctx::PoolShape need;
  need.dir = ctx->has_dir;
  need.lights = ctx->n_lights > 0;
  need.classes = ctx->n_classes < 1u ? 1u : ctx->n_classes;
  fh_ctx::PoolShape& have = ctx->pool_shape[slot];
  if (ctx->pool[slot].capacity >= capacity && (have.dir || !need.dir) && (have.lights || !need.lights) && have.classes >= need.classes) return FH_OK;
  if (ctx->pool[slot].capacity) {  // growing: nothing may still be running out of the old buffers
    FH_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2; ++k) FH_HIP(hipStreamSynchronize(ctx->aux_stream[k]));
    for (void* p : ctx->pool_allocs[slot]) (void)hipFree(p);
    ctx->pool_allocs[slot].clear();
    ctx->pool_alloc_bytes[slot] = 0;
    if (ctx->pool_target_by_caller && ctx->pool[slot].capacity > capacity) capacity = ctx->pool[slot].capacity;  // (a default-sized pool is re-made at the size the memory cap of this call allows)
    need.dir = need.dir || have.dir; need.lights = need.lights || have.lights; need.classes = need.classes > have.classes ? need.classes : have.classes;
    ctx->pool[slot] = PoolDev{};
    have = fh_ctx::PoolShape{};
    ctx->counters_in_flight[slot] = false;
  }
  PoolDev& P = ctx->pool[slot];
  auto alloc = [&](auto*& ptr, size_t count) -> hipError_t {
    void* raw = nullptr;
    hipError_t e = hipMalloc(&raw, count * sizeof(*ptr));
    if (e == hipSuccess) { ctx->pool_allocs[slot].push_back(raw); ctx->pool_alloc_bytes[slot] += count * sizeof(*ptr); ptr = (decltype(ptr))raw; }
    // FH_POISON=1 (tests): a new pool starts out full of 0xa5 instead of whatever the allocation held, so a kernel that reads a record or a queue entry nobody wrote shows at once
    if (e == hipSuccess && ctx->tun.poison_pools) e = hipMemsetAsync(raw, 0xa5, count * sizeof(*ptr), slot ? ctx->aux_stream[slot - 1] : ctx->stream);
    return e;
  };
  const size_t n = capacity;
  const uint32_t i_dir = 0u, i_sky = need.dir ? 1u : 0u, i_area = i_sky + 1u, i_light = i_sky + 1u + (need.lights ? 1u : 0u);
  const uint32_t sec_count = i_light + 1u;
  auto all = [&]() -> hipError_t {
    hipError_t e;
#define FH_POOL(ptr, count) if ((e = alloc(ptr, count)) != hipSuccess) return e
    float4 *state = nullptr, *aov = nullptr, *lp = nullptr;
    uint32_t* ident = nullptr;
    FH_POOL(state, n * 4); FH_POOL(P.rad, n); FH_POOL(ident, n * 2); FH_POOL(P.flags, n);
    FH_POOL(aov, n * 4); FH_POOL(P.sec, n * sec_count * 3);
    if (need.lights) FH_POOL(lp, n * 2);
    P.sec_count = sec_count;
    P.sec_index = i_dir | (i_sky << 8) | (i_area << 16) | (i_light << 24);
    P.ray_o.base = state; P.ray_d.base = state + 1; P.thr.base = state + 2; P.hit.base = state + 3;
    P.pixel.base = ident; P.nspp.base = ident + 1;
    P.aov_position.base = aov; P.aov_normal.base = aov + 1; P.aov_albedo.base = aov + 2; P.aov_texdepth.base = aov + 3;
    P.lp_a.base = lp; P.lp_b.base = lp ? lp + 1 : nullptr;
    FH_POOL(P.q_rad[0], n); FH_POOL(P.q_rad[1], n); FH_POOL(P.q_cls, n * need.classes); FH_POOL(P.q_sec, n); FH_POOL(P.q_prim, n);
    FH_POOL(P.counters, (size_t)kCounterStride * 66);  // up to 65 bounces per pass
    FH_POOL(P.key_sec, n); FH_POOL(P.key_rad, n); FH_POOL(P.q_tmp, n); FH_POOL(P.q_sec_sorted, n);
    FH_POOL(P.bins, (size_t)2 * kCells);
#undef FH_POOL
    // (on the stream this slot's passes run on, where the sorts that use the bins follow it: hipMemset on the null stream returns before the fill has run and is ordered
    // with NOTHING on a non-blocking stream -- a pass whose first sort overtook the fill counted into whatever the allocation held, turned that into scatter cursors and wrote
    // queue entries gigabytes away: the memory fault of profiles/README.md r4-10)
    return hipMemsetAsync(P.bins, 0, sizeof(uint32_t) * 2 * kCells, slot ? ctx->aux_stream[slot - 1] : ctx->stream);
  };
  const hipError_t e = all();
  if (e != hipSuccess) {  // leave the slot empty rather than half allocated: the next call starts from scratch
    for (void* p : ctx->pool_allocs[slot]) (void)hipFree(p);
    ctx->pool_allocs[slot].clear();
    ctx->pool_alloc_bytes[slot] = 0;
    P = PoolDev{};
    return fail(ctx, FH_E_HIP, std::string("path pool allocation: ") + hipGetErrorString(e));
  }
  P.capacity = capacity;
  have = need;
  return FH_OK;
}

// Dynamic LDS beyond the default limit has to be announced per kernel (hipFuncAttributeMaxDynamicSharedMemorySize).  Every kernel that keeps its stack in LDS
// is asked for its own static LDS (the fused tail carries 40 KB of tables next to the 14 KB of cooperative-test records the others have) and told the stack size when
// static + stack pass the default 64 KB; the largest static size is what fh_render checks against the CU's LDS.  Done once per BVH depth.
int configure_traversal_lds(fh_ctx* ctx, uint32_t stack_bytes)
{
  hipError_t err = hipSuccess;
  uint32_t max_static = kCoopLdsBytesPerBlock;
  auto set = [&](const void* fn) {
    hipFuncAttributes at{};
    if (hipFuncGetAttributes(&at, fn) == hipSuccess && (uint32_t)at.sharedSizeBytes > max_static) max_static = (uint32_t)at.sharedSizeBytes;
    if ((uint32_t)at.sharedSizeBytes + stack_bytes > 64u * 1024u) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stack_bytes);
      if (e != hipSuccess) err = e;
    }
  };
  for (int c = 0; c < 2; ++c)
    for (int a = 0; a < 2; ++a)
      with_bool(c != 0, [&](auto C) { with_bool(a != 0, [&](auto A) {
        set((const void*)k_trace_closest_stream<decltype(C)::value, decltype(A)::value>);
        set((const void*)k_trace_closest_coop<decltype(C)::value, decltype(A)::value>);
        for (int l = 0; l < 2; ++l)
          with_bool(l != 0, [&](auto Li) {
            set((const void*)k_trace_secondary_stream<decltype(C)::value, decltype(Li)::value, decltype(A)::value>);
            set((const void*)k_trace_merged_stream<decltype(Li)::value, decltype(A)::value>);
            set((const void*)k_trace_secondary_coop<decltype(C)::value, decltype(Li)::value, decltype(A)::value>);
            set((const void*)k_trace_secondary_static<decltype(C)::value, true, decltype(Li)::value, decltype(A)::value>);
            set((const void*)k_trace_secondary_static<decltype(C)::value, false, decltype(Li)::value, decltype(A)::value>);
          });
      }); });
  set((const void*)k_tail<L_DIFF>);
  set((const void*)k_tail<L_METAL | L_SPEC | L_DIFF>);
  set((const void*)k_tail<L_COAT | L_METAL | L_SPEC | L_DIFF>);
  set((const void*)k_tail<L_ALL>);
  if (err != hipSuccess) return fail(ctx, FH_E_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(err));
  ctx->lds_configured_bytes = stack_bytes;
  ctx->lds_static_max = max_static;
  return FH_OK;
}

// classify the owned pixels for this camera (k_split_pixels); cached until camera, resolution, ownership or scene bounds change
static int split_pixels(fh_ctx* ctx, const fh_camera* cam, const FrameDev& fr)
{
  float key[32] = {};
  for (int i = 0; i < 12; ++i) key[i] = cam->transform[i];
  key[12] = cam->fov; key[13] = cam->F; key[14] = cam->focus;
  for (int i = 0; i < 3; ++i) { key[15 + i] = ctx->scene_lo[i]; key[18 + i] = ctx->scene_hi[i]; }
  key[21] = (float)ctx->width; key[22] = (float)ctx->height; key[23] = (float)ctx->n_owned; key[24] = (float)ctx->shard_rank; key[25] = (float)ctx->shard_world;
  key[26] = (float)ctx->tile_w; key[27] = (float)ctx->tile_h;
  if (ctx->split_valid && std::memcmp(key, ctx->split_key, sizeof key) == 0) return FH_OK;
  ctx->split_valid = false;
  // the bound is derived for a rigid camera transform (rotation + translation): anything else renders every pixel through the passes
  const float* t = cam->transform;
  float dev = 0.0f;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const float d = t[a] * t[b] + t[4 + a] * t[4 + b] + t[8 + a] * t[8 + b];  // columns of the 3x3 block
      dev = fmaxf(dev, fabsf(d - (a == b ? 1.0f : 0.0f)));
    }
  std::memcpy(ctx->split_key, key, sizeof key);
  ctx->n_wave_px = ctx->n_owned; ctx->n_sky_px = 0;
  if (!(dev < 1e-4f)) { ctx->split_valid = true; return FH_OK; }  // (valid: "no sky pixels" for this key)
  if (ctx->split_capacity < ctx->n_owned) {
    FH_HIP(hipDeviceSynchronize());
    for (int k = 0; k < 4; ++k) { if (ctx->d_split[k]) (void)hipFree(ctx->d_split[k]); ctx->d_split[k] = nullptr; }
    ctx->split_capacity = 0;
    for (int k = 0; k < 4; ++k) FH_HIP(hipMalloc((void**)&ctx->d_split[k], 4ull * ctx->n_owned));
    ctx->split_capacity = ctx->n_owned;
  }
  // no launch of an earlier call may still read the lists that are rewritten here
  FH_HIP(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 2; ++k) FH_HIP(hipStreamSynchronize(ctx->aux_stream[k]));
  FH_HIP(hipStreamSynchronize(ctx->sky_stream));
  SplitDev sp{};
  const float c[3] = {0.5f * (ctx->scene_lo[0] + ctx->scene_hi[0]), 0.5f * (ctx->scene_lo[1] + ctx->scene_hi[1]), 0.5f * (ctx->scene_lo[2] + ctx->scene_hi[2])};
  const float w[3] = {c[0] - t[3], c[1] - t[7], c[2] - t[11]};
  sp.centre = mk3(t[0] * w[0] + t[4] * w[1] + t[8] * w[2], t[1] * w[0] + t[5] * w[1] + t[9] * w[2], t[2] * w[0] + t[6] * w[1] + t[10] * w[2]);  // R^T (C - T)
  const float e[3] = {ctx->scene_hi[0] - ctx->scene_lo[0], ctx->scene_hi[1] - ctx->scene_lo[1], ctx->scene_hi[2] - ctx->scene_lo[2]};
  sp.radius = 0.5f * sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  {
    const float f = fr.cam_inv_tan, a0w[3] = {t[2] * f + t[3], t[6] * f + t[7], t[10] * f + t[11]};  // the lens centre (0, 0, f) in world space
    float far2 = 0.0f;
    for (int k = 0; k < 8; ++k) {
      float d2 = 0.0f;
      for (int a = 0; a < 3; ++a) { const float c = ((k >> a) & 1) ? ctx->scene_hi[a] : ctx->scene_lo[a]; d2 += (c - a0w[a]) * (c - a0w[a]); }
      far2 = fmaxf(far2, d2);
    }
    sp.far = sqrtf(far2) * 1.001f;
  }
  sp.wave_px = ctx->d_split[0]; sp.wave_xy = ctx->d_split[1]; sp.sky_px = ctx->d_split[2]; sp.sky_xy = ctx->d_split[3];
  sp.counters = ctx->d_split_counters;
  FH_HIP(hipMemsetAsync(ctx->d_split_counters, 0, 8, ctx->stream));  // (the violation counter, word 2, keeps counting)
  hipLaunchKernelGGL(k_split_pixels, dim3(grid_for(ctx->n_owned)), dim3(kBlock), 0, ctx->stream, fr, sp, ctx->d_owned, ctx->d_owned_xy, ctx->n_owned);
  uint32_t n[2] = {0, 0};
  FH_HIP(hipMemcpyAsync(n, ctx->d_split_counters, 8, hipMemcpyDeviceToHost, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  if (n[0] + n[1] != ctx->n_owned) return fail(ctx, FH_E_HIP, "k_split_pixels lost pixels");
  ctx->n_wave_px = n[0]; ctx->n_sky_px = n[1];
  ctx->split_valid = true;
  if (getenv("FH_DEBUG_BVH")) fprintf(stderr, "[split] %u of %u owned pixels cannot see the scene (bounding sphere radius %.4f at camera-space (%.3f, %.3f, %.3f))\n", n[1], ctx->n_owned, sp.radius, sp.centre.x, sp.centre.y, sp.centre.z);
  return FH_OK;
}

int render_submit(fh_ctx* ctx, const fh_camera* cam, const float* bg, const fh_render_layers* layers, uint32_t n_samples, uint32_t max_depth, uint32_t seed)
{
  if (!ctx->scene_loaded || !ctx->bvh_valid) return fail(ctx, FH_E_INVALID, "fh_render: scene not uploaded or BVH not built");
  if (ctx->width == 0 || ctx->height == 0 || !ctx->d_sample_count) return fail(ctx, FH_E_INVALID, "fh_render: resolution not set");
  if (max_depth > 64) return fail(ctx, FH_E_INVALID, "fh_render: max_depth > 64 is not supported");
  if (ctx->n_owned == 0 || n_samples == 0) return FH_OK;
  // The default pool size (32 Mi paths per pool) is a wish: whenever a pool has to be allocated -- the first frame, after fh_scene_upload changed what a path record
  // holds, after a release -- all pools together are kept within A QUARTER of what the device has free at that moment, counting what the pools already hold as free.
  // (A pool is allocated for the paths a pass really starts -- the pixels that can see the scene x the samples of the pass, pool_ensure -- so the cap is an upper bound.)
  if (!ctx->pool_target_by_caller) {
    bool allocating = false;
    unsigned long long held = 0;
    for (int k = 0; k < (((ctx->flags & FH_FLAG_SERIAL_PASSES) != 0) ? 1 : ctx->n_slots); ++k) {
      const fh_ctx::PoolShape& have = ctx->pool_shape[k];
      if (ctx->pool[k].capacity == 0 || (ctx->has_dir && !have.dir) || (ctx->n_lights > 0 && !have.lights) || have.classes < (ctx->n_classes < 1u ? 1u : ctx->n_classes)) allocating = true;
      held += (unsigned long long)ctx->pool[k].capacity * pool_bytes_per_path(ctx);
    }
    if (allocating) {
      size_t free_b = 0, total_b = 0;
      ctx->pool_target = ctx->pool_target_default;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const unsigned long long cap = ((unsigned long long)free_b + held) / 4ull / ((unsigned long long)ctx->n_slots * pool_bytes_per_path(ctx));
        if (cap < ctx->pool_target) ctx->pool_target = cap > ctx->n_owned ? (uint32_t)cap : ctx->n_owned;
      }
      // (pools that exist already keep their size unless they have to be re-made: pool_ensure only grows)
    }
  }

  FrameDev fr{};
  fr.width = ctx->width; fr.height = ctx->height;
  fr.seed_hash = xxhash32(seed);
  fr.max_depth = max_depth;
  fr.has_dir = ctx->has_dir ? 1u : 0u;
  fr.has_hosek = ctx->has_hosek ? 1u : 0u;
  fr.has_ibl = ctx->d_ibl ? 1u : 0u;
  fr.ibl = fht_texture{nullptr, ctx->d_ibl, ctx->ibl_w, ctx->ibl_h, 0u};
  const uint32_t has_lights = ctx->n_lights > 0 ? 1u : 0u;
  fr.n1 = 3u + has_lights;
  fr.n2 = 3u + fr.has_dir + has_lights;
  for (int r = 0; r < 3; ++r) fr.cam_xf.r[r] = make_float4(cam->transform[4 * r], cam->transform[4 * r + 1], cam->transform[4 * r + 2], cam->transform[4 * r + 3]);
  fr.cam_inv_tan = 1.0f / tanf(0.5f * cam->fov);
  fr.cam_F = cam->F; fr.cam_focus = cam->focus;
  {  // camera.cu:33-36, in fp32 like the device code these lines used to be in (volatile: no contraction, no double-precision intermediates)
    volatile float f = fr.cam_inv_tan, b = cam->focus;
    volatile float inv_b = 1.0f / b;
    volatile float den = 1.0f + f;
    den = den - inv_b;
    volatile float a = 1.0f / den;
    volatile float apb = a + b;
    volatile float two_f = 2.0f * f;
    volatile float lr = two_f / cam->F;
    fr.cam_a_plus_b = apb;
    fr.cam_lens_radius = lr;
  }
  fr.bg = mk3(bg[0], bg[1], bg[2]);
  fr.sky_intensity = ctx->sky_intensity;
  fr.sun_dir = mk3(ctx->sun_dir[0], ctx->sun_dir[1], ctx->sun_dir[2]);
  fr.hosek = ctx->d_hosek;
  fr.dir_le = mk3(ctx->dir_le[0], ctx->dir_le[1], ctx->dir_le[2]);
  fr.dir_dir = mk3(ctx->dir_dir[0], ctx->dir_dir[1], ctx->dir_dir[2]);
  fr.dir_disk_radius = 1e9f * tanf(0.5f * ctx->dir_angle * kPi / 180.0f);
  fr.scene_lo = mk3(ctx->scene_lo[0], ctx->scene_lo[1], ctx->scene_lo[2]);
  fr.scene_hi = mk3(ctx->scene_hi[0], ctx->scene_hi[1], ctx->scene_hi[2]);
  {
    const float cells = (float)(1u << kCellBits);
    const float ex = ctx->scene_hi[0] - ctx->scene_lo[0], ey = ctx->scene_hi[1] - ctx->scene_lo[1], ez = ctx->scene_hi[2] - ctx->scene_lo[2];
    fr.cell_scale = mk3(ex > 0.0f ? cells / ex : 0.0f, ey > 0.0f ? cells / ey : 0.0f, ez > 0.0f ? cells / ez : 0.0f);
  }
  fr.sobol = ctx->d_sobol;
  fr.sobol_bytes = ctx->d_sobol_bytes;
  fr.lut.reflection = ctx->d_lut_refl;
  fr.lut.sheen = ctx->d_lut_sheen;

  LayersDev L{};
  L.beauty = (float4*)layers->beauty; L.position = (float4*)layers->position; L.depth = layers->depth;
  L.normal = (float4*)layers->normal; L.texcoord = (float4*)layers->texcoord; L.albedo = (float4*)layers->albedo;
  L.sample_count = ctx->d_sample_count;

  // ---- pixels that cannot see the scene are rendered by k_sky_pixels, the others by the passes below (see k_split_pixels).  Worth its fixed cost -- one small launch and a
  // host synchronisation whenever camera, resolution or scene bounds change -- for calls of many samples only: 2^27 camera paths, 64 spp of a 1080p frame
  const uint32_t* px_list = ctx->d_owned;
  const uint32_t* xy_list = ctx->d_owned_xy;
  uint32_t n_px = ctx->n_owned, n_sky = 0;
  const bool quirk_call = (ctx->flags & FH_FLAG_REFERENCE_FIRSTHIT) != 0 && n_samples > 1;
  if (ctx->tun.sky_split && !quirk_call && (unsigned long long)n_samples * ctx->n_owned >= (1ull << ctx->tun.sky_split_min_log2)) {
    const int rc = split_pixels(ctx, cam, fr);
    if (rc) return rc;
    if (ctx->split_valid && ctx->n_sky_px >= ctx->n_owned / 8u) {  // (a handful of sky pixels is not worth a second kernel)
      px_list = ctx->d_split[0]; xy_list = ctx->d_split[1];
      n_px = ctx->n_wave_px; n_sky = ctx->n_sky_px;
    }
  }
  uint32_t target = ctx->pool_target > ctx->n_owned ? ctx->pool_target : ctx->n_owned;
  uint32_t batch = n_px ? target / n_px : n_samples;
  if (batch > n_samples) batch = n_samples;
  if (batch > 65535u) batch = 65535u;  // (k_generate's grid has one row per sample of the pass)
  if (batch < 1) batch = 1;
  if (batch < n_samples) {  // equal passes, and whole rounds of the passes in flight: a call of 1024 samples with room for 248 per pass runs six passes of 171, not four of
    // 248 and a runt of 32 -- the last pass of an incomplete round has nothing to overlap with (configs[2]: 3 / 4 / 5 / 6 passes measure 7700 / 7577 / 7578 / 7598 Msamples/s, r5-5)
    uint32_t passes = (n_samples + batch - 1u) / batch;
    const uint32_t slots = (uint32_t)ctx->n_slots;
    if (slots > 1u && passes > slots && passes % slots != 0u && (passes / slots + 1u) * slots <= n_samples) passes = (passes / slots + 1u) * slots;
    batch = (n_samples + passes - 1u) / passes;
  }
  // (the passes of a call overlap, three in flight: a big call that would fit one or two passes is cut into three of the same size -- unless its passes run one after
  // the other anyway: the bug-compat mode and the measuring mode, where a split is pure overhead)
  if (!quirk_call && (ctx->flags & FH_FLAG_SERIAL_PASSES) == 0 && (n_samples + batch - 1u) / batch < (uint32_t)ctx->n_slots && n_samples >= (uint32_t)ctx->n_slots && (unsigned long long)n_samples * n_px >= 3ull << 24) batch = (n_samples + (uint32_t)ctx->n_slots - 1u) / (uint32_t)ctx->n_slots;

  const SceneDev sc_all = scene_dev(ctx);
  const SceneDev& sc = sc_all;  // (the passes below shadow this with their own copy: where rays start is a per-pass choice)
  const bool count = (ctx->flags & FH_FLAG_COUNT_TRAVERSAL) != 0;
  const bool clocks = (ctx->flags & FH_FLAG_TIME_KERNELS) != 0;
  TraceCounters tc_closest{ctx->d_trace_counters, ctx->d_trace_counters + 1, ctx->d_trace_counters + 2, ctx->d_trace_counters + 6, ctx->d_trace_counters + 7, ctx->d_trace_counters + 10,
                           clocks ? ctx->d_trace_counters + 27 : nullptr};
  TraceCounters tc_shadow{ctx->d_trace_counters + 3, ctx->d_trace_counters + 4, ctx->d_trace_counters + 5, ctx->d_trace_counters + 8, ctx->d_trace_counters + 9, ctx->d_trace_counters + 18,
                          clocks ? ctx->d_trace_counters + 29 : nullptr};

  // FH_FLAG_REFERENCE_FIRSTHIT with more than one sample per launch: per-pixel state carried through the launch, passes run one after the other
  const bool quirk = (ctx->flags & FH_FLAG_REFERENCE_FIRSTHIT) != 0 && n_samples > 1;
  if (quirk) {
    const size_t px = (size_t)ctx->width * ctx->height;
    if (ctx->quirk_pixels != px) {
      if (ctx->d_quirk_seen) (void)hipFree(ctx->d_quirk_seen);
      if (ctx->d_quirk_aov) (void)hipFree(ctx->d_quirk_aov);
      ctx->d_quirk_seen = nullptr; ctx->d_quirk_aov = nullptr; ctx->quirk_pixels = 0;
      FH_HIP(hipMalloc((void**)&ctx->d_quirk_seen, px * sizeof(uint32_t)));
      FH_HIP(hipMalloc((void**)&ctx->d_quirk_aov, px * 4 * sizeof(float4)));
      ctx->quirk_pixels = px;
    }
    FH_HIP(hipMemsetAsync(ctx->d_quirk_seen, 0, px * sizeof(uint32_t), ctx->stream));   // a launch starts with firsthit = true and a zeroed payload
    FH_HIP(hipMemsetAsync(ctx->d_quirk_aov, 0, px * 4 * sizeof(float4), ctx->stream));
  }
  if (!ctx->render_pending) { (void)hipEventRecord(ctx->ev_render_begin, ctx->stream); ctx->render_pending = true; }
  // whatever the caller queued on the main stream before this call (clears, uploads) comes first on the second stream too
  FH_HIP(hipEventRecord(ctx->ev_enter, ctx->stream));
  for (int k = 0; k + 1 < ctx->n_slots; ++k) FH_HIP(hipStreamWaitEvent(ctx->aux_stream[k], ctx->ev_enter, 0));
  int last_slot = 0;
  // (FH_FLAG_SERIAL_PASSES and FH_PIPELINE=0, the measuring modes: in line on the main stream, so that every kernel of the call is alone on the GPU, the spans add up and
  // a serial kernel trace shows the kernel's work instead of the time a starved background kernel was resident)
  hipStream_t sky_st = ((ctx->flags & FH_FLAG_SERIAL_PASSES) != 0 || ctx->n_slots == 1) ? ctx->stream : ctx->sky_stream;
  if (n_sky) {  // the sky pixels of this call, all samples at once, on a stream of their own next to the passes (they share no pixel with them)
    if (sky_st != ctx->stream) FH_HIP(hipStreamWaitEvent(sky_st, ctx->ev_enter, 0));
    Span sp(ctx, sky_st, 4);
    uint32_t sky_grid = grid_for(n_sky);
    if (ctx->tun.sky_blocks_per_cu && sky_grid > ctx->tun.n_cus * ctx->tun.sky_blocks_per_cu) sky_grid = ctx->tun.n_cus * ctx->tun.sky_blocks_per_cu;
    hipLaunchKernelGGL(k_sky_pixels, dim3(sky_grid), dim3(kBlock), 0, sky_st, fr, L, ctx->d_sample_issued, ctx->d_split[2], ctx->d_split[3], n_sky, n_samples, ctx->d_split_counters + 2);
    ctx->stats.paths += (uint64_t)n_sky * n_samples;
    ctx->stats.sky_pixel_samples += (uint64_t)n_sky * n_samples;
  }

  // device facts and developer switches were read once at fh_ctx_create (context.h: Tunables)
  const fh_ctx::Tunables& tun = ctx->tun;
  // wave-cooperative triangle tests (default for the wide BVH): queued candidates that trigger a round.  48 (r5-12) -- but ONE-PASS calls of scenes without cut-outs keep 32
  // (r5-13, profiles/r05_latency_defaults.log: the soup's fh_render(1 / 4 / 16) take 2.02 / 3.37 / 7.02 ms with 32 and 2.21 / 3.62 / 7.32 with 48 -- a launch of few rays
  // waits longer for 48 candidates; the interior with cut-outs is 1 % faster with 48 there too).  FH_COOP_T fixes it for every call.
  const uint32_t coop_flush = (!tun.coop_flush_fixed && batch >= n_samples && !sc.has_alpha && tun.coop_flush > 32u) ? 32u : tun.coop_flush;
  const bool coop = sc.use_bvh8 != 0 && sc.bvh8.n_tris < kCoopMaxTris && tun.coop;
  // streaming form (FH_STREAM=0: one fixed batch per wave).  Through a small tree every ray takes the same few steps: nothing to rebalance, and the fixed
  // batches run without the refill machinery (1000-triangle soup: closest 7.4 -> 4.6 ms, secondary 2.6 -> 1.0 ms per 256 spp; even at ~2 K nodes; behind at 20 K)
  const bool stream = coop && tun.stream && (tun.stream_forced || ctx->bvh8_n_nodes >= 4096u);
  // the traversal stack of every lane lives in LDS, one entry per level of the BVH (bvh_build.hip records the depth); the streaming kernels may spill deep levels (below)
  const uint32_t stack_bytes = lds_stack_bytes(stack_entries_for(ctx->bvh8_depth));
  const uint32_t cfg_bytes = lds_stack_bytes(stack_entries_for(ctx->bvh8_depth) + 1u);  // (+ 1: the anchor entry of rays that start below the root, fh_trace.h)
  if (sc.use_bvh8 && ctx->lds_configured_bytes != cfg_bytes) {  // kernels that may need more than the default 64 KB of LDS are told so once per BVH depth
    const int rc = configure_traversal_lds(ctx, cfg_bytes);
    if (rc) return rc;
  }
  if (sc.use_bvh8 && stack_bytes + ctx->lds_static_max > tun.lds_per_block) return fail(ctx, FH_E_UNSUPPORTED, "fh_render: BVH too deep for the LDS traversal stack");
  // all workgroups of a streaming launch are resident: as many per CU as its LDS (160 KB on gfx950) holds (at most 6: the kernels' register budget)
  const uint32_t stack_entries = stack_entries_for(ctx->bvh8_depth);
  // the streaming kernels' rays may start below the root: entry 0 of such a ray's stack is its anchor, the groups come on top (fh_trace.h: bottom-up start)
  const uint32_t stream_need = stack_entries + (sc.bvh8.parent ? 1u : 0u);
  // static LDS of a streaming kernel's workgroup: the cooperative-test records and, where candidates are parked for their any-hit test (AlphaDefer), the ring
  const uint32_t static_lds_closest = kCoopLdsBytesPerBlock + (sc.has_alpha && AlphaDefer<false, true>::value ? kAlphaLdsBytesPerBlock : 0u);
  const uint32_t static_lds_secondary = kCoopLdsBytesPerBlock + kTopLdsBytes + (sc.has_alpha && AlphaDefer<true, true>::value ? kAlphaLdsBytesPerBlock : 0u);
  if (stream) {  // what the runtime says really fits (LDS granularity, registers of the variant in use): a grid above it would leave blocks queued behind the resident ones
    const uint32_t key = stack_bytes | (count ? 1u : 0u) | (sc.has_alpha ? 2u : 0u) | (sc.n_lights > 0 ? 4u : 0u) | (sc.bvh8.parent ? 8u : 0u) | (tun.stack_lds_entries << 20);
    if (ctx->occupancy_key != key) {
      auto occupancy = [&](bool secondary, uint32_t entries) {
        const uint32_t bytes = lds_stack_bytes(entries);
        int r = 0;
        with_bool(count, [&](auto C) { with_bool(sc.has_alpha != 0, [&](auto A) {
          if (!secondary) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, k_trace_closest_stream<decltype(C)::value, decltype(A)::value>, kBlock, bytes);
          else with_bool(sc.n_lights > 0, [&](auto Li) {
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&r, k_trace_secondary_stream<decltype(C)::value, decltype(Li)::value, decltype(A)::value>, kBlock, bytes);
          });
        }); });
        return r;
      };
      // The stack costs LDS, and LDS costs resident workgroups.  Each of the two kernels keeps as many stack levels in LDS as still give it the workgroups per CU it is compiled
      // for (FH_STREAM_BLOCKS_CLOSEST / FH_STREAM_BLOCKS: at n workgroups per CU one has lds_per_cu / n bytes, and what the cooperative-test records leave is the stack's) and spills
      // the deeper levels, which are rarely reached, to global memory (GroupStack<true>).  FH_STACK_LDS=n fixes the number for both, FH_STACK_LDS=99 keeps everything in LDS.
      auto pick = [&](bool secondary, uint32_t blocks_wanted, uint32_t& entries_out, uint32_t& blocks_out) {
        const uint32_t lds_share = tun.lds_per_cu / blocks_wanted;
        const uint32_t static_lds = secondary ? static_lds_secondary : static_lds_closest;
        const uint32_t fit = lds_share > static_lds + 1280u ? (lds_share - static_lds) / 1280u : 1u;
        const uint32_t floor_entries = fit < 8u ? fit : 8u;
        uint32_t entries = stream_need;
        int got = occupancy(secondary, entries);
        if (tun.stack_lds_entries) entries = tun.stack_lds_entries < stream_need ? tun.stack_lds_entries : stream_need;
        else if (stream_need > floor_entries) {
          const int best = occupancy(secondary, floor_entries);
          while (entries > floor_entries && got < best) { --entries; got = occupancy(secondary, entries); }
        }
        if (entries != stream_need) got = occupancy(secondary, entries);
        entries_out = entries;
        blocks_out = got > 0 ? (uint32_t)got : 0u;
      };
      pick(false, FH_STREAM_BLOCKS_CLOSEST, ctx->stream_lds_entries, ctx->occupancy_blocks);
      pick(true, sc.n_lights > 0 ? FH_SECONDARY_BLOCKS_HEAVY : (sc.has_alpha ? FH_STREAM_BLOCKS_ALPHA : FH_STREAM_BLOCKS), ctx->stream_lds_entries_secondary, ctx->occupancy_blocks_secondary);
      ctx->occupancy_key = key;
      if (!count) {  // (what fh_kernel_info reports: the plan of the kernels that render, not of the instrumented variants a counting call has just been launched with)
        ctx->info_blocks[0] = ctx->occupancy_blocks; ctx->info_blocks[1] = ctx->occupancy_blocks_secondary;
        ctx->info_entries[0] = ctx->stream_lds_entries; ctx->info_entries[1] = ctx->stream_lds_entries_secondary;
      }
      if (getenv("FH_DEBUG_BVH"))
        fprintf(stderr, "[trace] stack of %u entries (%s); in LDS: closest %u (%u B + %u B per workgroup, %u resident workgroups per CU), secondary %u (%u B + %u B, %u workgroups)\n", stream_need, sc.bvh8.parent ? "rays start at the node of the face they leave" : "rays start at the root",
                ctx->stream_lds_entries, lds_stack_bytes(ctx->stream_lds_entries), static_lds_closest, ctx->occupancy_blocks, ctx->stream_lds_entries_secondary,
                lds_stack_bytes(ctx->stream_lds_entries_secondary), static_lds_secondary, ctx->occupancy_blocks_secondary);
    }
  }
  const uint32_t stream_entries = stream ? ctx->stream_lds_entries : stream_need, stream_entries_secondary = stream ? ctx->stream_lds_entries_secondary : stream_need;
  const uint32_t stream_stack_bytes = lds_stack_bytes(stream_entries), stream_stack_bytes_secondary = lds_stack_bytes(stream_entries_secondary);
  auto wgs_for = [&](uint32_t bytes, uint32_t static_lds, uint32_t compiled_for, uint32_t reported) {
    uint32_t w = tun.lds_per_cu / (bytes + static_lds);
    w = w > compiled_for ? compiled_for : (w < 1u ? 1u : w);
    if (stream && reported && reported < w) w = reported;
    return w;
  };
  const uint32_t wgs_closest = wgs_for(stream_stack_bytes, static_lds_closest, FH_STREAM_BLOCKS_CLOSEST, ctx->occupancy_blocks);
  const uint32_t wgs_secondary = wgs_for(stream_stack_bytes_secondary, static_lds_secondary, FH_STREAM_BLOCKS > FH_SECONDARY_BLOCKS_HEAVY ? FH_STREAM_BLOCKS : FH_SECONDARY_BLOCKS_HEAVY, ctx->occupancy_blocks_secondary);
  const uint32_t stream_grid = tun.n_cus * wgs_closest;
  const uint32_t stream_grid_secondary = tun.n_cus * wgs_secondary;
  // spill area of the streaming launches: [launch in flight: pass slot x (closest, secondary)][entry beyond the LDS part][thread of the launch]
  const uint32_t spill_entries = stream_entries < stream_need ? stream_need - stream_entries : 0u;
  const uint32_t spill_entries_secondary = stream_entries_secondary < stream_need ? stream_need - stream_entries_secondary : 0u;
  const size_t spill_threads = (size_t)(stream_grid > stream_grid_secondary ? stream_grid : stream_grid_secondary) * kBlock;
  const size_t spill_region = (size_t)(spill_entries > spill_entries_secondary ? spill_entries : spill_entries_secondary) * spill_threads;  // uint2 each
  if (spill_region * 6u > ctx->stack_spill_capacity) {
    FH_HIP(hipDeviceSynchronize());  // (launches of earlier calls may still use the old area)
    if (ctx->d_stack_spill) FH_HIP(hipFree(ctx->d_stack_spill));
    ctx->d_stack_spill = nullptr;
    ctx->stack_spill_capacity = 0;
    FH_HIP(hipMalloc((void**)&ctx->d_stack_spill, sizeof(uint2) * spill_region * 6u));
    ctx->stack_spill_capacity = spill_region * 6u;
  }
  // (rays through a small tree are cheap enough to run into the atomic rate of the work cursor: allow larger chunks there, stream_chunk_for)
  const uint32_t chunk_max = tun.stream_chunk_fixed ? tun.stream_chunk : (ctx->bvh8_n_nodes < 512u ? 256u : (ctx->bvh8_n_nodes < 4096u ? 128u : tun.stream_chunk));
  const uint32_t stream_refill = tun.stream_refill, stream_chunk = (tun.stream_chunk & 0xffffu) | ((chunk_max > tun.stream_chunk ? chunk_max : 0u) << 16);
  // (the closest-hit launch takes its own chunk size, FH_STREAM_CHUNK_CLOSEST)
  const uint32_t chunk_closest = tun.stream_chunk_closest ? tun.stream_chunk_closest : tun.stream_chunk;
  const uint32_t stream_chunk_closest = (chunk_closest & 0xffffu) | ((chunk_max > chunk_closest ? chunk_max : 0u) << 16);
  const uint32_t env_tail_depth = tun.tail_depth;
  // cell-ordered queues pay where rays of one cell share the nodes they fetch; a tree the fixed-batch kernels trace (under 4096 nodes) sits in the caches whatever
  // the order, and there the six sort launches per bounce are what a small frame waits for (Cornell box, 1 spp: 0.33 of 2.15 ms)
  const bool sort_queues = tun.sort_queues && (stream || tun.sort_small);
  const bool shade_three = tun.shade_wgs ? tun.shade_wgs == 3u : true;  // (above, FH_SHADE_BLOCKS)

  // (Small calls -- the reference's callers: 1 sample per call in the GUI, controller.cpp:224, 16 in rtcamp8, rtcamp8.cpp:183-189 -- cut into PIXEL sub-passes, each a pass of
  // its own in its own pool on its own stream, were built and measured in round 5: slower, configs[3] 1 spp 6.45 -> 8.24 ms.  tools/patches/r6_pruned_switches.patch, r5-4.)
  for (uint32_t done = 0; done < n_samples && n_px; done += batch) {
    const uint32_t nb = (n_samples - done) < batch ? (n_samples - done) : batch;
    const uint32_t n_paths = n_px * nb;
    const uint32_t grid = grid_for(n_paths);
    // n_slots passes in flight (three by default, FH_PIPELINE): pass j lives in pool j % n_slots on the stream of that slot.  Only two things order consecutive passes: the sample
    // indices (k_generate reads what k_bump_issued of the pass before wrote) and the running means (k_accumulate of pass j
    // follows k_accumulate of pass j - 1, so the floating-point result is that of a serial run)
    // (the bug-compat mode keeps every pass on the main stream: a pass needs the first-hit state the pass before it left)
    // FH_FLAG_SERIAL_PASSES does the same for measurements: kernels then run alone on the GPU and their HIP-event spans are kernel times
    const bool serial = quirk || (ctx->flags & FH_FLAG_SERIAL_PASSES) != 0;
    const int n_slots = serial ? 1 : ctx->n_slots;
    const int slot = serial ? 0 : (int)(ctx->pass_seq % (unsigned long long)n_slots), prev = serial ? ctx->last_slot_used : (slot + n_slots - 1) % n_slots;
    ctx->pass_seq++;
    ctx->stats.n_passes++;
    ctx->last_slot_used = slot;
    hipStream_t st = slot ? ctx->aux_stream[slot - 1] : ctx->stream;
    last_slot = slot;
    // A call that is ONE pass (the reference's callers: 1 or 16 samples per call) has no other pass to share the GPU with, and its bounces are chains of launches that
    // each end in a few long rays.  The secondary rays of bounce b and the closest-hit launch of bounce b + 1 do not depend on each other -- the secondary launch reads the
    // secondary-ray records and adds to the radiance, the closest-hit launch and the routing read rays and write hits -- so the secondary launch goes to a second stream;
    // the shade kernels of bounce b + 1, which overwrite the records it reads, wait for it.  (Calls of several passes overlap whole passes instead.)
    const bool single_pass = !serial && batch >= n_samples && n_samples == nb;
    // ... or, where the streaming kernels trace the scene, both in ONE launch (k_trace_merged_stream): one end instead of two, no second stream (FH_MERGE=0: the two-stream form)
    const bool merge = single_pass && stream && !count && tun.merge_trace;
    const bool overlap = single_pass && tun.overlap_secondary && !merge;
    hipStream_t sb = st;
    if (overlap) {
      sb = (st == ctx->aux_stream[0]) ? ctx->aux_stream[1] : ctx->aux_stream[0];
      while (ctx->ev_bounce.size() < 2u * (max_depth + 1u)) {
        hipEvent_t e = nullptr;
        FH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_bounce.push_back(e);
      }
    }
    // (The shade side of every bounce -- routing, shade kernels, queue sorts -- on a stream of its own, optionally of high priority, with the pass streams kept off some CUs,
    // was measured in round 4 and is slower: r4-6, tools/patches/r6_pruned_switches.patch.)
    hipStream_t sh = st;
    { const int rc = pool_ensure(ctx, slot, n_px * batch); if (rc) return rc; }
    const PoolDev& pool = ctx->pool[slot];
    FH_HIP(hipMemsetAsync(pool.counters, 0, sizeof(uint32_t) * kCounterStride * (max_depth + 1), st));
    // ---- where this pass's first-hit rays start their traversal (fh_trace.h: bottom-up start): forced by FH_BOTTOM_UP=0 / 1; else the first passes after a build alternate and
    // the device's own count of test rounds per shaded path decides (above): a dense scene gains (1M-triangle soup: -12 %), an interior of long rays does not (-0.4 %)
    int pass_bu = -1;  // -1: not a probing pass
    bool use_bu = false;
    if (sc_all.bvh8.parent && stream && (ctx->flags & FH_FLAG_ROOT_START) == 0u) {
      if (tun.bottom_up == 1u) use_bu = true;
      else if (tun.bottom_up == 2u) {
        // (only a pass whose secondary launch can climb is a probe: the merged launch of a one-pass call walks from the root whatever the pass says and counts no test
        // rounds, so counting it would fill one side of the comparison with cost 0 and fix the choice at "root" -- the reference's 1- and 16-sample calls come first in a GUI)
        if (ctx->bu_choice == 0 && !count && !merge && !sc_all.has_alpha) { use_bu = (ctx->bu_toggle++ & 1u) != 0u; pass_bu = use_bu ? 1 : 0; }
        else use_bu = ctx->bu_choice == 2;
      }
    }
    SceneDev sc = sc_all;
    if (!use_bu) { sc.bvh8.parent = nullptr; sc.face_node = nullptr; }
    if (prev != slot && ctx->gen_valid[prev]) FH_HIP(hipStreamWaitEvent(st, ctx->ev_gen[prev], 0));
    {
      Span sp(ctx, st, 4);
      hipLaunchKernelGGL(k_generate, dim3(grid_for((n_px + kGenChunks - 1) / kGenChunks), nb), dim3(kBlock), 0, st, fr, pool, ctx->d_sample_issued, px_list, xy_list, n_px);
      hipLaunchKernelGGL(k_bump_issued, dim3((n_px + kBlock - 1) / kBlock), dim3(kBlock), 0, st, ctx->d_sample_issued, px_list, n_px, nb);
      ctx->stats.n_generate_launches++;
    }
    FH_HIP(hipEventRecord(ctx->ev_gen[slot], st));
    ctx->gen_valid[slot] = true;
    ctx->stats.paths += n_paths;
    // bounces run as bounce-synchronous wavefront kernels; the survivors are finished by k_tail.  Adaptive mode picks the
    // first depth at which an earlier pass had at most kTailPaths survivors (counts come from an asynchronous snapshot of
    // the device counters: no host/device synchronisation)
    // (a wavefront bounce costs two traversal launches, each as long as its longest ray whatever the number of rays, and a dozen small launches: 0.5 ms on the
    // 1 M-triangle scene against 0.14 ms for a bounce inside the fused tail, so in a small pass the tail takes over earlier -- 1-spp 1080p frames: configs[2] 3.02 -> 2.73 ms
    // at 128 Ki, configs[1] 1.90 -> 1.77 ms at 256 Ki; in a big pass the tail runs next to the streaming launches of the passes in flight, two waves per SIMD
    // against their six, and more paths in it cost 0-1 %)
    const uint32_t kTailPaths = tun.tail_paths ? tun.tail_paths : (n_paths <= (1u << 22) ? 262144u : 65536u);
    for (int k = 0; k < 3; ++k) {
      if (!(ctx->counters_in_flight[k] && hipEventQuery(ctx->ev_counters[k]) == hipSuccess)) continue;
      ctx->counters_in_flight[k] = false;
      const uint32_t wd = ctx->counters_wave_depth[k];
      const uint32_t* hc = ctx->h_counters[k];
      if (ctx->counters_paths[k]) {  // the share of the pass's paths alive at each depth it ran as a wavefront (context.h: survival)
        const uint32_t n = wd < 65u ? wd : 65u;
        for (uint32_t d = 0; d <= n; ++d) ctx->survival[d] = (double)hc[d * kCounterStride + CNT_RAD] / (double)ctx->counters_paths[k];
        if (ctx->survival_n < n + 1u) ctx->survival_n = n + 1u;  // (a pass the tail took over early says nothing about the deeper shares: what an earlier pass of this scene measured stays)
      }
      if (ctx->counters_bu[k] >= 0 && ctx->bu_choice == 0) {  // (a pass of the scene's probing phase: what its secondary launches cost, by the issue model's weights)
        double cost = 0.0, items = 0.0;
        for (uint32_t d = 0; d < wd; ++d) { cost += 510.0 * hc[d * kCounterStride + CNT_COST_NODE] + 175.0 * hc[d * kCounterStride + CNT_COST_TRI]; items += hc[d * kCounterStride + CNT_SEC]; }
        ctx->bu_cost[ctx->counters_bu[k]] += cost; ctx->bu_items[ctx->counters_bu[k]] += items;
        if (ctx->bu_items[0] >= 65536.0 && ctx->bu_items[1] >= 65536.0) {
          const double off = ctx->bu_cost[0] / ctx->bu_items[0], on = ctx->bu_cost[1] / ctx->bu_items[1];
          ctx->bu_choice = on < 0.95 * off ? 2 : 1;
          if (getenv("FH_DEBUG_BVH")) fprintf(stderr, "[trace] rays that leave a surface start at %s: %.0f against %.0f SIMD cycles of tests per shaded path (from the root / from the face)\n", ctx->bu_choice == 2 ? "the node of their face" : "the root", off, on);
        }
      }
      ctx->counters_bu[k] = -1;
      if (tun.debug_tail) { fprintf(stderr, "[tail] slot %d wd %u paths %u survivors:", k, wd, ctx->counters_paths[k]); for (uint32_t d = 0; d <= wd; ++d) fprintf(stderr, " %u", ctx->h_counters[k][d * kCounterStride + CNT_RAD]); fprintf(stderr, "\n"); }
    }
    // the first depth at which THIS pass is expected to have at most kTailPaths survivors
    {
      uint32_t pick = 0;
      if (!ctx->survival_n) pick = n_paths > (1u << 22) ? max_depth : 2u;  // (nothing known: a big pass runs without the tail -- a wavefront bounce of few paths costs it 0.5 ms, a tail of millions of paths seconds)
      else {
        const uint32_t known = ctx->survival_n - 1u;
        for (uint32_t d = 1; d <= known && !pick; ++d)
          if ((double)n_paths * ctx->survival[d] <= (double)kTailPaths) pick = d;
        if (!pick) {
          // too many survivors even at the last depth counted: extrapolate with the survival ratio of the last bounce.  (Snapshots only
          // arrive when the host happens to be behind the GPU -- after a synchronisation -- so the depth has to be right in one step.)
          const double last = (double)n_paths * ctx->survival[known], prev = known ? (double)n_paths * ctx->survival[known - 1u] : 0.0;
          uint32_t more = 1;
          if (last > 0.0 && prev > last) {
            const double steps = ceil(log((double)kTailPaths / last) / log(last / prev));
            more = steps < 1.0 ? 1u : (steps > 16.0 ? 16u : (uint32_t)steps);
          }
          pick = known + more;
        }
      }
      ctx->auto_wave_depth = pick;
    }
    uint32_t wave_depth = ctx->tail_depth ? ctx->tail_depth : ctx->auto_wave_depth;
    if (env_tail_depth) wave_depth = env_tail_depth;
    if (wave_depth < 1u) wave_depth = 1u;
    if (wave_depth > max_depth) wave_depth = max_depth;
    PoolDev pd = pool, ps = pool;  // per-bounce views: pd rotates the radiance queues through the sorted buffers, ps reads the sorted secondary queue
    uint32_t* q_spare = pool.q_tmp;
    const uint32_t sort_blocks = n_paths / 16384u < 1u ? 1u : (n_paths / 16384u > 512u ? 512u : n_paths / 16384u);
    for (uint32_t depth = 0; depth < wave_depth; ++depth) {
      if (!(merge && depth > 0)) {  // (merged launches: the closest-hit rays of this bounce were traced next to the secondary rays of the bounce before)
        Span sp(ctx, st, 0);
        if (stream) {
          with_bool(count, [&](auto C) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_closest_stream<decltype(C)::value, decltype(A)::value>), dim3(grid < stream_grid ? grid : stream_grid), dim3(kBlock), stream_stack_bytes, st, sc, pd, depth, tc_closest,
                               coop_flush, stream_refill, stream_chunk_closest, tun.stream_min_rays, StackSpill{spill_entries ? ctx->d_stack_spill + (size_t)(2 * slot) * spill_region : nullptr, stream_entries});
          }); });
        } else if (coop) {
          with_bool(count, [&](auto C) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_closest_coop<decltype(C)::value, decltype(A)::value>), dim3(grid), dim3(kBlock), stack_bytes, st, sc, pd, depth, tc_closest, coop_flush);
          }); });
        } else {
          with_bool(count, [&](auto C) { with_bool(sc.use_bvh8 != 0, [&](auto W) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_closest_static<decltype(C)::value, decltype(W)::value, decltype(A)::value>), dim3(grid), dim3(kBlock), 0, st, sc, pd, depth, tc_closest);
          }); }); });
        }
        ctx->stats.n_closest_launches++;
      }
      if (quirk && depth == 0) hipLaunchKernelGGL(k_firsthit_scan, dim3(grid_for(n_px)), dim3(kBlock), 0, st, pd, px_list, n_px, nb, ctx->d_quirk_seen);
      {
        Span sp(ctx, sh, 6);
        hipLaunchKernelGGL(k_route, dim3(grid), dim3(kBlock), 0, sh, sc, pd, depth, ctx->n_classes, count ? ctx->d_trace_counters + 26 : nullptr);
      }
      if (overlap && depth > 0) FH_HIP(hipStreamWaitEvent(st, ctx->ev_bounce[2u * (depth - 1u) + 1u], 0));  // the secondary launch of the bounce before reads what the shade kernels and the sorts below overwrite
      {
        Span sp(ctx, sh, 2);
        for (uint32_t c = 0; c < ctx->n_classes; ++c) dispatch_shade(sh, grid, ctx->class_lobes[c], sc, fr, pd, c, depth, shade_three);
        if (depth == 0) hipLaunchKernelGGL(k_miss_primary, dim3(grid), dim3(kBlock), 0, sh, fr, pd);
        ctx->stats.n_shade_launches += ctx->n_classes;
      }
      {
        Span sp(ctx, sh, 6);
        // secondary rays of this bounce and the radiance rays of the next one, each into cell order.  ONE-PASS calls (r6-7): the order buys their launches nothing -- every
        // line they touch is touched for the first time whatever the order -- and the six sort launches per bounce are pure chain: 4- and 16-sample calls of configs[3] are
        // 3 % / 2 % faster without them; only the queue the fused tail takes over keeps its order (FH_SORT_ONEPASS=1: all, =0: none)
        const bool tail_next = depth + 1u == wave_depth && wave_depth < max_depth;
        const bool sort_sec = sort_queues && (!single_pass || tun.sort_onepass == 1u);
        const bool sort_rad = sort_queues && (!single_pass || tun.sort_onepass == 1u || (tun.sort_onepass == 2u && tail_next));
        ps = pd;
        if (sort_sec) {
          sort_queue_by_cell(sh, sort_blocks, pool.counters + depth * kCounterStride + CNT_SEC, pool.q_sec, pool.key_sec, pool.bins, pool.q_sec_sorted);
          ps.q_sec = pool.q_sec_sorted;
        }
        if (sort_rad) {
          const uint32_t nxt = (depth + 1u) & 1u;
          sort_queue_by_cell(sh, sort_blocks, pool.counters + (depth + 1u) * kCounterStride + CNT_RAD, pd.q_rad[nxt], pool.key_rad, pool.bins, q_spare);
          uint32_t* const unsorted = pd.q_rad[nxt];
          pd.q_rad[nxt] = q_spare;  // the next bounce reads the sorted queue ...
          q_spare = unsorted;       // ... and the buffer it came from is the next scratch target
        }
      }
      if (merge && depth + 1u < wave_depth) {
        Span sp(ctx, st, 1);
        with_bool(sc.n_lights > 0, [&](auto Li) { with_bool(sc.has_alpha != 0, [&](auto A) {
          hipLaunchKernelGGL((k_trace_merged_stream<decltype(Li)::value, decltype(A)::value>), dim3(grid < stream_grid_secondary ? grid : stream_grid_secondary), dim3(kBlock), stream_stack_bytes_secondary, st, sc,
                             fr, ps, pd, depth, tc_shadow, coop_flush, stream_refill, stream_chunk, tun.stream_min_rays,
                             StackSpill{spill_entries_secondary ? ctx->d_stack_spill + (size_t)(2 * slot + 1) * spill_region : nullptr, stream_entries_secondary, pass_bu >= 0 ? 1u : 0u});
        }); });
        ctx->stats.n_shadow_launches++;
        ctx->stats.n_closest_launches++;
      } else {
        if (overlap) { FH_HIP(hipEventRecord(ctx->ev_bounce[2u * depth], st)); FH_HIP(hipStreamWaitEvent(sb, ctx->ev_bounce[2u * depth], 0)); }
        Span sp(ctx, sb, 1);
        if (stream) {
          with_bool(count, [&](auto C) { with_bool(sc.n_lights > 0, [&](auto Li) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_secondary_stream<decltype(C)::value, decltype(Li)::value, decltype(A)::value>), dim3(grid < stream_grid_secondary ? grid : stream_grid_secondary), dim3(kBlock), stream_stack_bytes_secondary, sb, sc,
                               fr, ps, depth, tc_shadow, coop_flush, stream_refill, stream_chunk, tun.stream_min_rays,
                               StackSpill{spill_entries_secondary ? ctx->d_stack_spill + (size_t)(2 * slot + 1) * spill_region : nullptr, stream_entries_secondary, pass_bu >= 0 ? 1u : 0u});
          }); }); });
        } else if (coop) {
          with_bool(count, [&](auto C) { with_bool(sc.n_lights > 0, [&](auto Li) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_secondary_coop<decltype(C)::value, decltype(Li)::value, decltype(A)::value>), dim3(grid), dim3(kBlock), stack_bytes, sb, sc, fr, ps, depth, tc_shadow,
                               coop_flush);
          }); }); });
        } else {
          with_bool(count, [&](auto C) { with_bool(sc.use_bvh8 != 0, [&](auto W) { with_bool(sc.n_lights > 0, [&](auto Li) { with_bool(sc.has_alpha != 0, [&](auto A) {
            hipLaunchKernelGGL((k_trace_secondary_static<decltype(C)::value, decltype(W)::value, decltype(Li)::value, decltype(A)::value>), dim3(grid), dim3(kBlock), stack_bytes, sb, sc, fr, ps,
                               depth, tc_shadow);
          }); }); }); });
        }
        ctx->stats.n_shadow_launches++;
      }
      if (overlap) FH_HIP(hipEventRecord(ctx->ev_bounce[2u * depth + 1u], sb));
    }
    if (overlap && wave_depth > 0) FH_HIP(hipStreamWaitEvent(st, ctx->ev_bounce[2u * (wave_depth - 1u) + 1u], 0));  // the tail and the accumulate read the radiance the last secondary launch completes
    if (wave_depth < max_depth) {
      Span sp(ctx, st, 3);
      uint32_t lobes = 0;  // every lobe a material of the scene can have: the tail shades all classes in one kernel
      for (uint32_t c = 0; c < ctx->n_classes; ++c) lobes |= ctx->class_lobes[c];
      const dim3 tg(grid_for(n_paths / 16 + 1)), tb(kBlock);
      const uint32_t tl = sc.use_bvh8 ? stack_bytes : 0u, tf = coop ? coop_flush : 0u;
      if ((lobes & ~(uint32_t)L_DIFF) == 0) hipLaunchKernelGGL(k_tail<L_DIFF>, tg, tb, tl, st, sc, fr, pd, wave_depth, tf);
      else if ((lobes & ~(uint32_t)(L_METAL | L_SPEC | L_DIFF)) == 0) hipLaunchKernelGGL((k_tail<L_METAL | L_SPEC | L_DIFF>), tg, tb, tl, st, sc, fr, pd, wave_depth, tf);
      else if ((lobes & ~(uint32_t)(L_COAT | L_METAL | L_SPEC | L_DIFF)) == 0) hipLaunchKernelGGL((k_tail<L_COAT | L_METAL | L_SPEC | L_DIFF>), tg, tb, tl, st, sc, fr, pd, wave_depth, tf);
      else hipLaunchKernelGGL(k_tail<L_ALL>, tg, tb, tl, st, sc, fr, pd, wave_depth, tf);
      ctx->stats.n_tail_launches++;
    }
    if (prev != slot && ctx->acc_valid[prev]) FH_HIP(hipStreamWaitEvent(st, ctx->ev_acc[prev], 0));
    {
      Span sp(ctx, st, 5);
      if (quirk) hipLaunchKernelGGL(k_accumulate<true>, dim3(grid_for(n_px)), dim3(kBlock), 0, st, pool, L, px_list, n_px, nb, ctx->d_quirk_aov);
      else hipLaunchKernelGGL(k_accumulate<false>, dim3(grid_for(n_px)), dim3(kBlock), 0, st, pool, L, px_list, n_px, nb, (float4*)nullptr);
      ctx->stats.n_accumulate_launches++;
    }
    FH_HIP(hipEventRecord(ctx->ev_acc[slot], st));
    ctx->acc_valid[slot] = true;
    if (!ctx->counters_in_flight[slot] && ctx->h_counters[slot]) {
      FH_HIP(hipMemcpyAsync(ctx->h_counters[slot], pool.counters, sizeof(uint32_t) * kCounterStride * (max_depth + 1), hipMemcpyDeviceToHost, st));
      FH_HIP(hipEventRecord(ctx->ev_counters[slot], st));
      ctx->counters_in_flight[slot] = true;
      ctx->counters_wave_depth[slot] = wave_depth;
      ctx->counters_paths[slot] = n_paths;
      ctx->counters_bu[slot] = pass_bu;
    }
  }
  // join: later work on the main stream (pack, post-process, copies, the caller's clears) sees every pass of this call
  // (the accumulates form a chain across the streams, so the last one implies all the others)
  if (last_slot != 0 && n_px) FH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_acc[last_slot], 0));
  if (n_sky && sky_st != ctx->stream) { FH_HIP(hipEventRecord(ctx->ev_sky, sky_st)); FH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_sky, 0)); }
  FH_HIP(hipGetLastError());
  return FH_OK;
}

}  // namespace fh
