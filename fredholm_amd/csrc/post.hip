// post.hip -- bloom / chromatic aberration / exposure / Uchimura tone map / sRGB.
//
// Replaces post_process_kernel_launch (fredholm/kernels/src/post-process.cu:5-35) and its kernels
// bloom_kernel_0 (:60-74), bloom_kernel_1 (:76-109), tone_mapping_kernel (:111-153), copy_kernel (:49-58),
// helpers fredholm/kernels/include/kernels/post-process.h:13-124.
// Kept quirks: the launch grid is floor(W/16) x floor(H/16) blocks of 16x16, so rows/columns past the
// last full block are never written (:9-11); the blur weight is exp(-d^2 / (2*bloom_sigma)) (:97-98);
// the chromatic-aberration offset is divided by W*H (:124-125).
// The 33x33 blur reads its 48x48 source tile through LDS once per block instead of 1089 global
// loads per pixel; taps are summed in the reference's (v outer, u inner) order so results match
// the checker bit for bit.
#include <hip/hip_runtime.h>

#include <vector>

#include "context.h"
#include "fh_tonemap.h"

namespace fh {
namespace {

constexpr int kR = 16;            // blur radius
constexpr int kT = 16;            // tile edge
constexpr int kS = kT + 2 * kR;   // staged edge (48)

__global__ void __launch_bounds__(256) k_bloom_threshold(const float4* in, int w, int gw, int gh, float threshold, float4* out)
{
  const int i = blockIdx.x * kT + threadIdx.x, j = blockIdx.y * kT + threadIdx.y;
  if (i >= gw || j >= gh) return;
  const float4 b = in[i + w * j];
  const float l = b.x * 0.2126729f + b.y * 0.7151522f + b.z * 0.0721750f;
  out[i + w * j] = l > threshold ? b : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

__global__ void __launch_bounds__(256) k_bloom_blur(const float4* in, const float4* hi, int w, int h, const float* weights, float wsum, float4* out)
{
  __shared__ float4 tile[kS * kS];
  __shared__ float wt[33 * 33];
  const int tid = threadIdx.y * kT + threadIdx.x;
  const int x0 = blockIdx.x * kT - kR, y0 = blockIdx.y * kT - kR;
  for (int k = tid; k < kS * kS; k += 256) {
    int x = x0 + k % kS, y = y0 + k / kS;
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    tile[k] = hi[x + w * y];
  }
  for (int k = tid; k < 33 * 33; k += 256) wt[k] = weights[k];
  __syncthreads();
  const int i = blockIdx.x * kT + threadIdx.x, j = blockIdx.y * kT + threadIdx.y;
  float sx = 0.0f, sy = 0.0f, sz = 0.0f, sw = 0.0f;
  for (int v = 0; v < 33; ++v)
    for (int u = 0; u < 33; ++u) {
      const float hh = wt[v * 33 + u];
      const float4 b1 = tile[(threadIdx.y + v) * kS + threadIdx.x + u];
      sx += hh * b1.x; sy += hh * b1.y; sz += hh * b1.z; sw += hh * b1.w;
    }
  const float inv = 1.0f / wsum;
  const float4 b0 = in[i + w * j];
  out[i + w * j] = make_float4(b0.x + sx * inv, b0.y + sy * inv, b0.z + sz * inv, b0.w + sw * inv);
}

__global__ void __launch_bounds__(256) k_copy(const float4* in, int w, int gw, int gh, float4* out)
{
  const int i = blockIdx.x * kT + threadIdx.x, j = blockIdx.y * kT + threadIdx.y;
  if (i >= gw || j >= gh) return;
  out[i + w * j] = in[i + w * j];
}

__global__ void __launch_bounds__(256) k_tone_map(const float4* in, int w, int h, int gw, int gh, float exposure, float ca, float4* out)
{
  const int i = blockIdx.x * kT + threadIdx.x, j = blockIdx.y * kT + threadIdx.y;
  if (i >= gw || j >= gh) return;
  const float uvx = (float)i / w, uvy = (float)j / h;
  const float inv = 1.0f / (float)(w * h);
  const float dx = (uvx - 0.5f) * inv * ca, dy = (uvy - 0.5f) * inv * ca;
  const float rx = clamp01f(uvx - 0.0f * dx), ry = clamp01f(uvy - 0.0f * dy);
  const float gx = clamp01f(uvx - 1.0f * dx), gy = clamp01f(uvy - 1.0f * dy);
  const float bx = clamp01f(uvx - 2.0f * dx), by = clamp01f(uvy - 2.0f * dy);
  const int ir = (int)(rx * w + w * (ry * h)), ig = (int)(gx * w + w * (gy * h)), ib = (int)(bx * w + w * (by * h));
  float r = in[ir].x, g = in[ig].y, b = in[ib].z;
  r *= exposure; g *= exposure; b *= exposure;
  r = srgb1(uchimura1(r)); g = srgb1(uchimura1(g)); b = srgb1(uchimura1(b));
  out[i + w * j] = make_float4(r, g, b, 1.0f);
}


// ------------------------------------------------------------------------------------------------
// Denoiser slot.  The reference calls NVIDIA's OptiX AI denoiser (fredholm/include/fredholm/denoiser.h:14-146: HDR model, albedo + normal
// guides, optional 2x upscale) -- a proprietary network with no counterpart here and nothing to be bit-compatible with.  What fills the slot is
// the classic guided filter it replaced in many renderers: the edge-avoiding a-trous wavelet filter (Dammertz, Sewtz, Hanika, Lensch, HPG 2010)
// on albedo-demodulated radiance, five passes of a 5x5 B3-spline kernel with hole sizes 1, 2, 4, 8, 16, edge-stopping weights on colour
// (relative to the local level, sigma halved per pass), normal and albedo.  Same inputs and output as the reference's class; `upscale` doubles the output by pixel replication.
constexpr float kDnSigmaColor = 2.0f, kDnSigmaNormal = 0.35f, kDnSigmaAlbedo = 0.2f, kDnAlbedoFloor = 0.01f;  // colour: relative to the mean of the two irradiances
constexpr int kDnPasses = 5;

__device__ __forceinline__ float dn_floor(float a) { return fmaxf(a, kDnAlbedoFloor); }
__device__ __forceinline__ float dn_finite(float v) { return (v != v || fabsf(v) > 3.0e38f) ? 0.0f : v; }

// pass `it`: src = irradiance of the previous pass (pass 0 reads beauty and demodulates); the last pass re-modulates and writes alpha 1
__global__ void __launch_bounds__(256) k_atrous(const float4* src, const float4* beauty, const float4* normal, const float4* albedo, int w, int h, int it, int last, int upscale, float4* dst)
{
  const int x = blockIdx.x * 16 + threadIdx.x, y = blockIdx.y * 16 + threadIdx.y;
  if (x >= w || y >= h) return;
  const int step = 1 << it;
  const float kern[3] = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
  const float inv_sc = 1.0f / (kDnSigmaColor * kDnSigmaColor * (1.0f / (float)(1 << (2 * it)))), inv_sn = 1.0f / (kDnSigmaNormal * kDnSigmaNormal),
              inv_sa = 1.0f / (kDnSigmaAlbedo * kDnSigmaAlbedo);
  auto irradiance = [&](int i) -> float4 {
    if (it != 0) return src[i];
    const float4 b = beauty[i], a = albedo[i];
    return make_float4(dn_finite(b.x) / dn_floor(a.x), dn_finite(b.y) / dn_floor(a.y), dn_finite(b.z) / dn_floor(a.z), 0.0f);
  };
  const int p = x + w * y;
  const float4 cp = irradiance(p), np = normal[p], ap = albedo[p];
  float sx = 0.0f, sy = 0.0f, sz = 0.0f, sw = 0.0f;
  for (int dy = -2; dy <= 2; ++dy)
    for (int dx = -2; dx <= 2; ++dx) {
      int qx = x + dx * step, qy = y + dy * step;
      qx = qx < 0 ? 0 : (qx > w - 1 ? w - 1 : qx);
      qy = qy < 0 ? 0 : (qy > h - 1 ? h - 1 : qy);
      const int q = qx + w * qy;
      const float4 cq = irradiance(q), nq = normal[q], aq = albedo[q];
      const float dcx = cq.x - cp.x, dcy = cq.y - cp.y, dcz = cq.z - cp.z;
      const float dnx = nq.x - np.x, dny = nq.y - np.y, dnz = nq.z - np.z;
      const float dax = aq.x - ap.x, day = aq.y - ap.y, daz = aq.z - ap.z;
      const float m = (cq.x + cq.y + cq.z) + (cp.x + cp.y + cp.z);  // colour distance relative to the local level: HDR input
      const float den = m * m * (1.0f / 9.0f) + 1e-4f;
      const float e = ((dcx * dcx + dcy * dcy + dcz * dcz) / den) * inv_sc + (dnx * dnx + dny * dny + dnz * dnz) * inv_sn + (dax * dax + day * day + daz * daz) * inv_sa;
      const float wgt = kern[dx < 0 ? -dx : dx] * kern[dy < 0 ? -dy : dy] * fhe_exp(-e);
      sx += wgt * cq.x; sy += wgt * cq.y; sz += wgt * cq.z; sw += wgt;
    }
  const float inv = 1.0f / sw;  // the centre tap has weight 9/64: never zero
  float4 o = make_float4(sx * inv, sy * inv, sz * inv, 0.0f);
  if (!last) { dst[p] = o; return; }
  o = make_float4(o.x * dn_floor(ap.x), o.y * dn_floor(ap.y), o.z * dn_floor(ap.z), 1.0f);
  if (!upscale) { dst[p] = o; return; }
  const int w2 = 2 * w;
  dst[2 * x + w2 * (2 * y)] = o; dst[2 * x + 1 + w2 * (2 * y)] = o; dst[2 * x + w2 * (2 * y + 1)] = o; dst[2 * x + 1 + w2 * (2 * y + 1)] = o;
}

}  // namespace

int post_process_submit(fh_ctx* ctx, const float* in, float* hi, float* tmp, int w, int h, const fh_post_params* pp, float* out)
{
  hipStream_t st = ctx->stream;
  // FH_FLAG_TIME_KERNELS: HIP events around the whole chain (fh_stats.post_ms, reduced at fh_sync like the render spans: kind 7)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  const bool timed = (ctx->flags & FH_FLAG_TIME_KERNELS) != 0;
  auto take = [&]() { hipEvent_t e = nullptr; if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
  if (timed) { ev_a = take(); ev_b = take(); (void)hipEventRecord(ev_a, st); }
  const int bx = w / kT > 1 ? w / kT : 1, by = h / kT > 1 ? h / kT : 1;  // floor division, post-process.cu:9-11
  const int gw = bx * kT < w ? bx * kT : w, gh = by * kT < h ? by * kT : h;
  const dim3 grid(bx, by), block(kT, kT);
  if (pp->use_bloom) {
    if (w < kT || h < kT) return fail(ctx, FH_E_UNSUPPORTED, "bloom needs an image of at least 16x16 pixels");
    // the 1089 weights depend on bloom_sigma only: computed and uploaded when sigma changes, kept in the context otherwise
    // (no allocation, copy or host synchronisation per frame)
    if (!ctx->d_bloom_weights) FH_HIP(hipMalloc((void**)&ctx->d_bloom_weights, 33 * 33 * sizeof(float)));
    if (ctx->bloom_sigma_cached != pp->bloom_sigma) {
      std::vector<float> wt(33 * 33);
      float wsum = 0.0f;
      for (int v = -kR; v <= kR; ++v)
        for (int u = -kR; u <= kR; ++u) {
          const float dist2 = (float)(u * u + v * v);
          const float hh = fhe_exp(-dist2 / (2.0f * pp->bloom_sigma));
          wt[(v + kR) * 33 + (u + kR)] = hh;
          wsum += hh;
        }
      FH_HIP(hipStreamSynchronize(st));  // an earlier frame may still read the old weights
      FH_HIP(hipMemcpy(ctx->d_bloom_weights, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
      ctx->bloom_sigma_cached = pp->bloom_sigma;
      ctx->bloom_wsum = wsum;
    }
    hipLaunchKernelGGL(k_bloom_threshold, grid, block, 0, st, (const float4*)in, w, gw, gh, pp->bloom_threshold, (float4*)hi);
    hipLaunchKernelGGL(k_bloom_blur, grid, block, 0, st, (const float4*)in, (const float4*)hi, w, h, ctx->d_bloom_weights, ctx->bloom_wsum, (float4*)tmp);
  } else {
    hipLaunchKernelGGL(k_copy, grid, block, 0, st, (const float4*)in, w, gw, gh, (float4*)tmp);
  }
  const float exposure = exposure_from_ev100(ev100_of(1.0f, 1.0f, pp->ISO));
  hipLaunchKernelGGL(k_tone_map, grid, block, 0, st, (const float4*)tmp, w, h, gw, gh, exposure, pp->chromatic_aberration, (float4*)out);
  if (timed) { (void)hipEventRecord(ev_b, st); ctx->spans.push_back({ev_a, ev_b, 7}); ctx->stats.n_post_launches++; }
  FH_HIP(hipGetLastError());
  return FH_OK;
}

int denoise_submit(fh_ctx* ctx, int w, int h, const float* beauty, const float* normal, const float* albedo, float* out, int upscale)
{
  hipStream_t st = ctx->stream;
  const size_t px = (size_t)w * h;
  if (ctx->denoise_pixels < px) {  // two ping-pong irradiance buffers, kept in the context
    for (int k = 0; k < 2; ++k) { if (ctx->d_denoise_tmp[k]) (void)hipFree(ctx->d_denoise_tmp[k]); ctx->d_denoise_tmp[k] = nullptr; }
    ctx->denoise_pixels = 0;
    for (int k = 0; k < 2; ++k) FH_HIP(hipMalloc((void**)&ctx->d_denoise_tmp[k], px * sizeof(float4)));
    ctx->denoise_pixels = px;
  }
  const dim3 grid((w + 15) / 16, (h + 15) / 16), block(16, 16);
  const float4* src = nullptr;
  for (int it = 0; it < kDnPasses; ++it) {
    const int last = it == kDnPasses - 1;
    float4* dst = last ? (float4*)out : ctx->d_denoise_tmp[it & 1];
    hipLaunchKernelGGL(k_atrous, grid, block, 0, st, src, (const float4*)beauty, (const float4*)normal, (const float4*)albedo, w, h, it, last, upscale, dst);
    src = dst;
  }
  FH_HIP(hipGetLastError());
  return FH_OK;
}

}  // namespace fh
