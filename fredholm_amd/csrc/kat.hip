// kat.hip -- batch entry points that evaluate, ON THE DEVICE, the same device functions the render
// kernels use (hashes, samplers, warps, BSDF, sky, camera, traversal).  They exist so the parity
// tests can compare each building block with the CPU checker through the C ABI; they are not on
// the render path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/fredholm_hip_test.h"
#include "context.h"
#include "fh_bsdf.h"
#include "fh_tonemap.h"
#include "fh_trace.h"

namespace fh {
namespace {

template <typename T>
struct Tmp {
  T* p = nullptr;
  size_t n = 0;
  ~Tmp() { if (p) (void)hipFree(p); }
  hipError_t up(const T* host, size_t count)
  {
    n = count;
    hipError_t e = hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (host && count) e = hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
  hipError_t down(T* host) { return hipMemcpy(host, p, n * sizeof(T), hipMemcpyDeviceToHost); }
};

__global__ void k_hash(int kind, uint32_t n, const uint32_t* in, uint32_t* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  out[i] = kind == 0 ? xxhash32(a) : kind == 1 ? xxhash32(a, b, c) : kind == 2 ? xxhash32(a, b, c, d) : cmj_permute(a, b, c);
}
__global__ void k_cmj(uint32_t n, const uint32_t* in, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f2 r = cmj_draw(in[4 * i], in[4 * i + 1], in[4 * i + 2], xxhash32(in[4 * i + 3]));
  // the block form k_sky_pixels draws with (fh_sampler.h: cmj_block) has to give the same bits: a draw on which the two differ comes back as NaN
  const f2 rb = cmj_draw_in_block(cmj_block(in[4 * i] / 16u, in[4 * i + 1], in[4 * i + 2], xxhash32(in[4 * i + 3])), in[4 * i]);
  const bool same = __float_as_uint(r.x) == __float_as_uint(rb.x) && __float_as_uint(r.y) == __float_as_uint(rb.y);
  out[2 * i] = same ? r.x : __uint_as_float(0x7fc00000u);
  out[2 * i + 1] = same ? r.y : __uint_as_float(0x7fc00000u);
}
__global__ void k_sobol(uint32_t n, const uint32_t* in, const uint32_t* table, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t dim = in[4 * i + 1];
  out[i] = sobol_draw(table + (dim & 1023u) * 52u, in[4 * i], dim, in[4 * i + 2]);
}
__global__ void k_elementary(int fn, uint32_t n, const float* x, const float* y, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = 0.0f;
  switch (fn) {
    case 0: r = fhe_sin(x[i]); break;
    case 1: r = fhe_cos(x[i]); break;
    case 2: r = fhe_exp(x[i]); break;
    case 3: r = fhe_log(x[i]); break;
    case 4: r = fhe_pow(x[i], y[i]); break;
    case 5: r = fhe_acos(x[i]); break;
    case 6: r = fhe_atan2(x[i], y[i]); break;
    case 7: r = fhe_log2(x[i]); break;
  }
  out[i] = r;
}
__global__ void k_warp(int kind, uint32_t n, const float* u, const float* wo, float ax, float ay, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f2 uu = mk2(u[2 * i], u[2 * i + 1]);
  if (kind == 0) { const f2 r = concentric_disk(uu); out[2 * i] = r.x; out[2 * i + 1] = r.y; }
  else if (kind == 1) { const f3 r = cosine_hemisphere(uu); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
  else if (kind == 2) { const f2 r = triangle_barycentric(uu); out[2 * i] = r.x; out[2 * i + 1] = r.y; }
  else { const f3 r = sample_vndf(mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), ax, ay, uu); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
}

FH_D MatParams params_of(const MaterialDev& m)
{
  MatParams p;
  p.diffuse = m.w[0]; p.base_color = mk3(m.w[1], m.w[2], m.w[3]); p.diffuse_roughness = m.w[5];
  p.specular = m.w[6]; p.specular_color = mk3(m.w[7], m.w[8], m.w[9]); p.specular_roughness = clampf(m.w[11], 0.01f, 1.0f);
  p.metalness = m.w[13];
  p.coat = clampf(m.w[16], 0.0f, 1.0f); p.coat_color = mk3(1.0f); p.coat_roughness = clampf(m.w[21], 0.0f, 1.0f);
  p.transmission = m.w[23]; p.transmission_color = mk3(m.w[24], m.w[25], m.w[26]);
  p.sheen = m.w[27]; p.sheen_color = mk3(m.w[28], m.w[29], m.w[30]); p.sheen_roughness = m.w[31];
  p.subsurface = m.w[32]; p.subsurface_color = mk3(m.w[33], m.w[34], m.w[35]);
  p.thin_walled = m.w[36];
  return p;
}

template <uint32_t LOBES>
__global__ void k_bsdf(MaterialDev mat, int entering, float eta_given, BsdfTables lut, uint32_t n, const float* wo, const float* wi, const float* u1, const float* u2, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 o = mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), in = mk3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
  Bsdf<LOBES> b;
  b.init(o, params_of(mat), entering != 0, lut);
  if (eta_given > 0.0f) { b.ni = 1.0f; b.nt = eta_given; b.eta = eta_given; }  // fh_kat_bsdf_ior: the lobes at a relative index other than the constructor's
  const f3 e = b.eval(o, in);
  f3 f;
  float pdf;
  const f3 s = b.sample(o, u1[i], mk2(u2[2 * i], u2[2 * i + 1]), f, pdf);
  float* r = out + 18 * i;
  r[0] = e.x; r[1] = e.y; r[2] = e.z; r[3] = b.eval_pdf(o, in);
  r[4] = s.x; r[5] = s.y; r[6] = s.z; r[7] = f.x; r[8] = f.y; r[9] = f.z; r[10] = pdf;
  for (int k = 0; k < 7; ++k) r[11 + k] = b.pmf(k);
}

// small math building blocks that have a reference-built counterpart (oracle/_ref/libref_lut_math_post.so): albedo LUT fetches
// (lut.cu:957-1081), shading-frame helpers (math.cu:7-35, :90-118) and the tone-map helper chain (post-process.h:13-124)
__global__ void k_math(int kind, BsdfTables lut, uint32_t n, const float* in, uint32_t si, float* out, uint32_t so)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = in + (size_t)si * i;
  float* o = out + (size_t)so * i;
  switch (kind) {
    case FH_MATH_ALBEDO_REFLECTION: o[0] = lut_reflection_albedo(lut.reflection, mk3(0.0f, a[0], 0.0f), a[1], a[2]); break;
    case FH_MATH_ALBEDO_SHEEN: o[0] = lut_sheen_albedo(lut.sheen, mk3(0.0f, a[0], 0.0f), a[1]); break;
    case FH_MATH_ONB: { f3 t, b; onb(mk3(a[0], a[1], a[2]), t, b); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = b.x; o[4] = b.y; o[5] = b.z; break; }
    case FH_MATH_TO_LOCAL: { const f3 r = to_local(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(a[6], a[7], a[8]), mk3(a[9], a[10], a[11])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
    case FH_MATH_TO_WORLD: { const f3 r = to_world(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(a[6], a[7], a[8]), mk3(a[9], a[10], a[11])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
    case FH_MATH_SPHERICAL: {  // cartesian_to_spherical as env_radiance evaluates it for the IBL lookup
      o[0] = fhe_acos(clampf(a[1], -1.0f, 1.0f));
      float phi = fhe_atan2(a[2], a[0]);
      if (phi < 0) phi += 2.0f * kPi;
      o[1] = phi;
      break;
    }
    case FH_MATH_LUMINANCE: o[0] = lum(mk3(a[0], a[1], a[2])); break;
    case FH_MATH_UCHIMURA: o[0] = uchimura1(a[0]); o[1] = uchimura1(a[1]); o[2] = uchimura1(a[2]); break;
    case FH_MATH_LINEAR_TO_SRGB: o[0] = srgb1(a[0]); o[1] = srgb1(a[1]); o[2] = srgb1(a[2]); break;
    case FH_MATH_EXPOSURE: o[0] = ev100_of(a[0], a[1], a[2]); o[1] = exposure_from_ev100(o[0]); break;
    case FH_MATH_TONE_MAP_TAIL: {
      const float e = exposure_from_ev100(ev100_of(1.0f, 1.0f, a[3]));
      o[0] = srgb1(uchimura1(a[0] * e)); o[1] = srgb1(uchimura1(a[1] * e)); o[2] = srgb1(uchimura1(a[2] * e));
      break;
    }
    case FH_MATH_POST_LUMINANCE: o[0] = luminance_rgb(a[0], a[1], a[2]); break;
  }
}

// measured HBM bandwidth of this GPU: the denominator SURVEY.md 8(d) asks the roofline fraction to be quoted against.  A streaming
// float4 read (every lane 16 B per load, grid-stride, four loads in flight) and a float4 copy over buffers far larger than the 256 MiB
// Infinity Cache.
__global__ void __launch_bounds__(256) k_bw_read(const float4* src, size_t n, float* sink)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4 a = make_float4(0, 0, 0, 0), b = a, c = a, d = a;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w; b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
    c.x += v2.x; c.y += v2.y; c.z += v2.z; c.w += v2.w; d.x += v3.x; d.y += v3.y; d.z += v3.z; d.w += v3.w;
  }
  for (; i < n; i += stride) { const float4 v = src[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  const float r = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
  if (r == 123.456f) sink[0] = r;  // never true for the zero-filled source; keeps the loads alive
}
__global__ void __launch_bounds__(256) k_bw_copy(const float4* src, float4* dst, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// the product's texture unit (include/fh_texture_unit.h) on the device, for comparison with the checker's independent texture unit
__global__ void k_tex2d(fht_texture tex, const float* srgb_lut, uint32_t n, const float* uv, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[4];
  fht_tex2d(&tex, srgb_lut, uv[2 * i], uv[2 * i + 1], o);
  out[4 * i] = o[0]; out[4 * i + 1] = o[1]; out[4 * i + 2] = o[2]; out[4 * i + 3] = o[3];
}

__global__ void k_sky(HosekSky st, f3 sun, float intensity, uint32_t n, const float* d, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 r = hosek_radiance(st, sun, intensity, mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
  out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

__global__ void k_camera(m34 xf, float inv_tan, float F, float focus, uint32_t width, uint32_t height, uint32_t seed_hash, uint32_t n, const uint32_t* pix, const uint32_t* nspp, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t image_idx = pix[i], n_spp = nspp[i];
  const uint32_t px = image_idx % width, py = image_idx / width;
  f2 u = cmj_draw(n_spp, image_idx, 0u, seed_hash);
  float uvx = (2.0f * (px + u.x) - width) / height;
  const float uvy = (2.0f * (py + u.y) - height) / height;
  uvx = -uvx;
  u = cmj_draw(n_spp, image_idx, 1u, seed_hash);
  const float f = inv_tan, b = focus;
  const float a = 1.0f / (1.0f + f - 1.0f / b);
  const float lens_radius = 2.0f * f / F;
  const f3 p_sensor = mk3(uvx, uvy, 0.0f), p_lens_center = mk3(0.0f, 0.0f, f);
  const f2 pd = lens_radius * concentric_disk(u);
  const f3 p_lens = p_lens_center + mk3(pd.x, pd.y, 0.0f);
  const f3 s2c = normalize(p_lens_center - p_sensor);
  const f3 p_object = p_sensor + ((a + b) / s2c.z) * s2c;
  const f3 org = xform_point(xf, p_lens);
  f3 d = normalize(p_object - p_lens);
  d.z *= -1.0f;
  const f3 dir = xform_dir(xf, d);
  out[6 * i] = org.x; out[6 * i + 1] = org.y; out[6 * i + 2] = org.z; out[6 * i + 3] = dir.x; out[6 * i + 4] = dir.y; out[6 * i + 5] = dir.z;
}

__global__ void k_offset(uint32_t n, const float* p, const float* nn, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 r = offset_origin(mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), mk3(nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]));
  out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}

__global__ void k_trace_batch(SceneDev sc, uint32_t n, const float* rays7, int any_hit, float* tuv, uint32_t* prim)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rays7 + 7 * (size_t)i;
  HitRec h;
  uint32_t a = 0, b = 0;
  const bool ok = any_hit ? traverse<true, false>(sc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], h, a, b)
                          : traverse<false, false>(sc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], h, a, b);
  tuv[3 * i] = ok ? h.t : 0.0f; tuv[3 * i + 1] = ok ? h.u : 0.0f; tuv[3 * i + 2] = ok ? h.v : 0.0f;
  prim[i] = ok ? h.prim : 0xffffffffu;
}

// same batch through the wave-cooperative traversal the render kernels use (all 64 lanes of a wave enter together)
template <bool ANY, bool ALPHA>
__global__ void __launch_bounds__(256) k_trace_batch_coop(SceneDev sc, uint32_t n, const float* rays7, float* tuv, uint32_t* prim, uint32_t flush)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kCoopLdsBytesPerWave];
  const CoopLds cl = coop_lds(lds, threadIdx.x >> 6);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const float* r = rays7 + 7 * (size_t)(valid ? i : 0u);
  HitRec h;
  uint32_t a = 0, b = 0;
  const bool ok = traverse_bvh8_coop<ANY, false, false, ALPHA>(sc.bvh8, valid, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], h, a, b, nullptr, cl, flush, nullptr, 0, &sc);
  if (!valid) return;
  tuv[3 * i] = ok ? h.t : 0.0f; tuv[3 * i + 1] = ok ? h.u : 0.0f; tuv[3 * i + 2] = ok ? h.v : 0.0f;
  prim[i] = ok ? h.prim : 0xffffffffu;
}

// fhe_sqrt (the short device sequence, include/fh_elementary.h) against the compiler's IEEE sqrtf over EVERY float bit pattern; optionally the results
// of `n_sample` given inputs are returned for a comparison with the host's sqrtf
__global__ void k_sqrt_all(unsigned long long* bad)
{
  const uint32_t stride = gridDim.x * blockDim.x;  // 2^22: 1024 iterations cover the 2^32 patterns
  unsigned long long n = 0;
  uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t it = 0; it < 1024u; ++it, u += stride) {
    const float x = __uint_as_float(u), a = fhe_sqrt(x), b = sqrtf(x);
    if (!(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b))) n++;
  }
  if (n) atomicAdd(bad, n);
}
__global__ void k_sqrt_some(uint32_t n, const float* in, float* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fhe_sqrt(in[i]);
}

uint32_t blocks(uint32_t n) { return (n + 255) / 256; }

}  // namespace
}  // namespace fh

using namespace fh;

#define KCTX(ctx)                  \
  if (!(ctx)) return FH_E_INVALID; \
  if (hipSetDevice((ctx)->device) != hipSuccess) return fh::fail(ctx, FH_E_HIP, "hipSetDevice failed")

extern "C" {

int fh_trace_rays(fh_ctx* ctx, uint32_t n, const float* rays7, int any_hit, float* tuv, uint32_t* prim)
{
  KCTX(ctx);
  if (!ctx->scene_loaded || !ctx->bvh_valid) return fail(ctx, FH_E_INVALID, "fh_trace_rays: scene/BVH missing");
  if (n == 0) return FH_OK;
  Tmp<float> r, t;
  Tmp<uint32_t> p;
  FH_HIP(r.up(rays7, 7ull * n)); FH_HIP(t.up(nullptr, 3ull * n)); FH_HIP(p.up(nullptr, n));
  const SceneDev sd = scene_dev(ctx);
  uint32_t flush = 32u;
  flush = ctx->tun.coop_flush;
  if (sd.use_bvh8 && sd.bvh8.n_tris < kCoopMaxTris && ctx->tun.coop) {
    const dim3 g(blocks(n)), b(256);
    if (any_hit && sd.has_alpha) hipLaunchKernelGGL((k_trace_batch_coop<true, true>), g, b, 0, ctx->stream, sd, n, r.p, t.p, p.p, flush);
    else if (any_hit) hipLaunchKernelGGL((k_trace_batch_coop<true, false>), g, b, 0, ctx->stream, sd, n, r.p, t.p, p.p, flush);
    else if (sd.has_alpha) hipLaunchKernelGGL((k_trace_batch_coop<false, true>), g, b, 0, ctx->stream, sd, n, r.p, t.p, p.p, flush);
    else hipLaunchKernelGGL((k_trace_batch_coop<false, false>), g, b, 0, ctx->stream, sd, n, r.p, t.p, p.p, flush);
  } else {
    hipLaunchKernelGGL(k_trace_batch, dim3(blocks(n)), dim3(256), 0, ctx->stream, sd, n, r.p, any_hit, t.p, p.p);
  }
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(t.down(tuv)); FH_HIP(p.down(prim));
  return FH_OK;
}

int fh_kat_hash(fh_ctx* ctx, int kind, uint32_t n, const uint32_t* in4, uint32_t* out)
{
  KCTX(ctx);
  Tmp<uint32_t> a, o;
  FH_HIP(a.up(in4, 4ull * n)); FH_HIP(o.up(nullptr, n));
  hipLaunchKernelGGL(k_hash, dim3(blocks(n)), dim3(256), 0, ctx->stream, kind, n, a.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out));
  return FH_OK;
}
int fh_kat_cmj(fh_ctx* ctx, uint32_t n, const uint32_t* in4, float* out2)
{
  KCTX(ctx);
  Tmp<uint32_t> a;
  Tmp<float> o;
  FH_HIP(a.up(in4, 4ull * n)); FH_HIP(o.up(nullptr, 2ull * n));
  hipLaunchKernelGGL(k_cmj, dim3(blocks(n)), dim3(256), 0, ctx->stream, n, a.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out2));
  return FH_OK;
}
int fh_kat_sobol(fh_ctx* ctx, uint32_t n, const uint32_t* in4, float* out)
{
  KCTX(ctx);
  Tmp<uint32_t> a;
  Tmp<float> o;
  FH_HIP(a.up(in4, 4ull * n)); FH_HIP(o.up(nullptr, n));
  hipLaunchKernelGGL(k_sobol, dim3(blocks(n)), dim3(256), 0, ctx->stream, n, a.p, ctx->d_sobol, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out));
  return FH_OK;
}
int fh_kat_elementary(fh_ctx* ctx, int fn, uint32_t n, const float* x, const float* y, float* out)
{
  KCTX(ctx);
  Tmp<float> a, b, o;
  FH_HIP(a.up(x, n)); FH_HIP(b.up(y ? y : x, n)); FH_HIP(o.up(nullptr, n));
  hipLaunchKernelGGL(k_elementary, dim3(blocks(n)), dim3(256), 0, ctx->stream, fn, n, a.p, b.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out));
  return FH_OK;
}
int fh_kat_warp(fh_ctx* ctx, int kind, uint32_t n, const float* u2, const float* wo3, const float* alpha2, float* out)
{
  KCTX(ctx);
  const uint32_t width = (kind == 1 || kind == 3) ? 3u : 2u;
  Tmp<float> u, w, o;
  FH_HIP(u.up(u2, 2ull * n)); FH_HIP(w.up(wo3, wo3 ? 3ull * n : 0)); FH_HIP(o.up(nullptr, (size_t)width * n));
  hipLaunchKernelGGL(k_warp, dim3(blocks(n)), dim3(256), 0, ctx->stream, kind, n, u.p, w.p, alpha2 ? alpha2[0] : 1.0f, alpha2 ? alpha2[1] : 1.0f, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out));
  return FH_OK;
}
static int kat_bsdf(fh_ctx* ctx, const fh_material* material, int entering, float eta_given, uint32_t lobes_mask, uint32_t n, const float* wo3, const float* wi3, const float* u1, const float* u2, float* out18)
{
  KCTX(ctx);
  if (!material) return FH_E_INVALID;
  MaterialDev m{};
  std::memcpy(m.w, material, 180);
  const BsdfTables lut{ctx->d_lut_refl, ctx->d_lut_sheen};
  Tmp<float> a, b, c, d, o;
  FH_HIP(a.up(wo3, 3ull * n)); FH_HIP(b.up(wi3, 3ull * n)); FH_HIP(c.up(u1, n)); FH_HIP(d.up(u2, 2ull * n)); FH_HIP(o.up(nullptr, 18ull * n));
  const dim3 g(blocks(n)), t(256);
  switch (lobes_mask) {
    case L_DIFF: hipLaunchKernelGGL(k_bsdf<L_DIFF>, g, t, 0, ctx->stream, m, entering, eta_given, lut, n, a.p, b.p, c.p, d.p, o.p); break;
    case L_METAL: hipLaunchKernelGGL(k_bsdf<L_METAL>, g, t, 0, ctx->stream, m, entering, eta_given, lut, n, a.p, b.p, c.p, d.p, o.p); break;
    case L_SPEC | L_DIFF: hipLaunchKernelGGL(k_bsdf<L_SPEC | L_DIFF>, g, t, 0, ctx->stream, m, entering, eta_given, lut, n, a.p, b.p, c.p, d.p, o.p); break;
    case L_METAL | L_SPEC | L_DIFF: hipLaunchKernelGGL(k_bsdf<L_METAL | L_SPEC | L_DIFF>, g, t, 0, ctx->stream, m, entering, eta_given, lut, n, a.p, b.p, c.p, d.p, o.p); break;
    default: hipLaunchKernelGGL(k_bsdf<L_ALL>, g, t, 0, ctx->stream, m, entering, eta_given, lut, n, a.p, b.p, c.p, d.p, o.p); break;
  }
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out18));
  return FH_OK;
}
int fh_kat_bsdf(fh_ctx* ctx, const fh_material* material, int entering, uint32_t lobes_mask, uint32_t n, const float* wo3, const float* wi3, const float* u1, const float* u2, float* out18)
{
  return kat_bsdf(ctx, material, entering, 0.0f, lobes_mask, n, wo3, wi3, u1, u2, out18);
}
int fh_kat_bsdf_ior(fh_ctx* ctx, const fh_material* material, float eta, uint32_t lobes_mask, uint32_t n, const float* wo3, const float* wi3, const float* u1, const float* u2, float* out18)
{
  if (!(eta > 0.0f)) return FH_E_INVALID;
  return kat_bsdf(ctx, material, 1, eta, lobes_mask, n, wo3, wi3, u1, u2, out18);
}
int fh_kat_sky(fh_ctx* ctx, uint32_t n, const float* dirs3, float* out3)
{
  KCTX(ctx);
  Tmp<float> a, o;
  FH_HIP(a.up(dirs3, 3ull * n)); FH_HIP(o.up(nullptr, 3ull * n));
  hipLaunchKernelGGL(k_sky, dim3(blocks(n)), dim3(256), 0, ctx->stream, ctx->hosek, mk3(ctx->sun_dir[0], ctx->sun_dir[1], ctx->sun_dir[2]), ctx->sky_intensity, n, a.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out3));
  return FH_OK;
}
int fh_kat_hosek_state(fh_ctx* ctx, float* out30)
{
  KCTX(ctx);
  std::memcpy(out30, &ctx->hosek, sizeof ctx->hosek);
  return FH_OK;
}
int fh_kat_camera(fh_ctx* ctx, const fh_camera* cam, uint32_t width, uint32_t height, uint32_t seed, uint32_t n, const uint32_t* pixel_idx, const uint32_t* n_spp, float* out6)
{
  KCTX(ctx);
  m34 xf;
  for (int r = 0; r < 3; ++r) xf.r[r] = make_float4(cam->transform[4 * r], cam->transform[4 * r + 1], cam->transform[4 * r + 2], cam->transform[4 * r + 3]);
  Tmp<uint32_t> a, b;
  Tmp<float> o;
  FH_HIP(a.up(pixel_idx, n)); FH_HIP(b.up(n_spp, n)); FH_HIP(o.up(nullptr, 6ull * n));
  hipLaunchKernelGGL(k_camera, dim3(blocks(n)), dim3(256), 0, ctx->stream, xf, 1.0f / tanf(0.5f * cam->fov), cam->F, cam->focus, width, height, xxhash32(seed), n, a.p, b.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out6));
  return FH_OK;
}
int fh_kat_offset_origin(fh_ctx* ctx, uint32_t n, const float* p3, const float* n3, float* out3)
{
  KCTX(ctx);
  Tmp<float> a, b, o;
  FH_HIP(a.up(p3, 3ull * n)); FH_HIP(b.up(n3, 3ull * n)); FH_HIP(o.up(nullptr, 3ull * n));
  hipLaunchKernelGGL(k_offset, dim3(blocks(n)), dim3(256), 0, ctx->stream, n, a.p, b.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out3));
  return FH_OK;
}
int fh_kat_sqrt(fh_ctx* ctx, unsigned long long* mismatches_over_all_inputs, uint32_t n_sample, const float* sample_in, float* sample_out)
{
  KCTX(ctx);
  if (!mismatches_over_all_inputs || (n_sample && (!sample_in || !sample_out))) return fail(ctx, FH_E_INVALID, "fh_kat_sqrt: null argument");
  Tmp<unsigned long long> bad;
  const unsigned long long zero = 0;
  FH_HIP(bad.up(&zero, 1));
  hipLaunchKernelGGL(k_sqrt_all, dim3(1u << 14), dim3(256), 0, ctx->stream, bad.p);
  Tmp<float> a, o;
  if (n_sample) {
    FH_HIP(a.up(sample_in, n_sample)); FH_HIP(o.up(nullptr, n_sample));
    hipLaunchKernelGGL(k_sqrt_some, dim3(blocks(n_sample)), dim3(256), 0, ctx->stream, n_sample, a.p, o.p);
  }
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(bad.down(mismatches_over_all_inputs));
  if (n_sample) FH_HIP(o.down(sample_out));
  return FH_OK;
}
int fh_kat_math(fh_ctx* ctx, int kind, uint32_t n, const float* in, float* out)
{
  KCTX(ctx);
  static const uint32_t widths[FH_MATH_COUNT][2] = {{3, 1}, {2, 1}, {3, 6}, {12, 3}, {12, 3}, {3, 2}, {3, 1}, {3, 3}, {3, 3}, {3, 2}, {4, 3}, {3, 1}};
  if (kind < 0 || kind >= FH_MATH_COUNT) return fail(ctx, FH_E_INVALID, "fh_kat_math: unknown kind");
  const uint32_t si = widths[kind][0], so = widths[kind][1];
  Tmp<float> a, o;
  FH_HIP(a.up(in, (size_t)si * n)); FH_HIP(o.up(nullptr, (size_t)so * n));
  BsdfTables lut;
  lut.reflection = ctx->d_lut_refl;
  lut.sheen = ctx->d_lut_sheen;
  hipLaunchKernelGGL(k_math, dim3(blocks(n)), dim3(256), 0, ctx->stream, kind, lut, n, a.p, si, o.p, so);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out));
  return FH_OK;
}
int fh_measure_bandwidth(fh_ctx* ctx, uint64_t bytes, uint32_t iters, double* read_gbs, double* copy_gbs)
{
  KCTX(ctx);
  if (!read_gbs || !copy_gbs || bytes < (1ull << 20) || iters == 0) return fail(ctx, FH_E_INVALID, "fh_measure_bandwidth: bad argument");
  const size_t n = (size_t)(bytes / 16);
  Tmp<float4> a, b;
  Tmp<float> sink;
  FH_HIP(a.up(nullptr, n)); FH_HIP(b.up(nullptr, n)); FH_HIP(sink.up(nullptr, 1));
  FH_HIP(hipMemsetAsync(a.p, 0, n * 16, ctx->stream));
  FH_HIP(hipMemsetAsync(b.p, 0, n * 16, ctx->stream));
  const dim3 block(256);
  *read_gbs = *copy_gbs = 0.0;
  for (uint32_t per_cu : {8u, 16u, 32u, 64u}) {  // the best of a few grid sizes: the right number of workgroups in flight is a property of the GPU, not of the renderer
    const dim3 grid(ctx->tun.n_cus * per_cu);
    hipEvent_t e0, e1, e2;
    FH_HIP(hipEventCreate(&e0)); FH_HIP(hipEventCreate(&e1)); FH_HIP(hipEventCreate(&e2));
    hipLaunchKernelGGL(k_bw_read, grid, block, 0, ctx->stream, (const float4*)a.p, n, sink.p);  // warm-up (page tables, clocks)
    hipLaunchKernelGGL(k_bw_copy, grid, block, 0, ctx->stream, (const float4*)a.p, b.p, n);
    FH_HIP(hipEventRecord(e0, ctx->stream));
    for (uint32_t k = 0; k < iters; ++k) hipLaunchKernelGGL(k_bw_read, grid, block, 0, ctx->stream, (const float4*)((k & 1u) ? b.p : a.p), n, sink.p);
    FH_HIP(hipEventRecord(e1, ctx->stream));
    for (uint32_t k = 0; k < iters; ++k) hipLaunchKernelGGL(k_bw_copy, grid, block, 0, ctx->stream, (const float4*)((k & 1u) ? b.p : a.p), (k & 1u) ? a.p : b.p, n);
    FH_HIP(hipEventRecord(e2, ctx->stream));
    FH_HIP(hipStreamSynchronize(ctx->stream));
    float ms_r = 0.0f, ms_c = 0.0f;
    FH_HIP(hipEventElapsedTime(&ms_r, e0, e1));
    FH_HIP(hipEventElapsedTime(&ms_c, e1, e2));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    const double r = (double)n * 16.0 * iters / (ms_r * 1e-3) / 1e9, c = 2.0 * (double)n * 16.0 * iters / (ms_c * 1e-3) / 1e9;  // copy: bytes read + bytes written
    if (r > *read_gbs) *read_gbs = r;
    if (c > *copy_gbs) *copy_gbs = c;
  }
  // the runtime's own device-to-device copy of the same buffers: the better of the two is reported as the copy bandwidth
  hipEvent_t m0, m1;
  FH_HIP(hipEventCreate(&m0)); FH_HIP(hipEventCreate(&m1));
  FH_HIP(hipEventRecord(m0, ctx->stream));
  for (uint32_t k = 0; k < iters; ++k) FH_HIP(hipMemcpyAsync((k & 1u) ? (void*)a.p : (void*)b.p, (k & 1u) ? (const void*)b.p : (const void*)a.p, n * 16, hipMemcpyDeviceToDevice, ctx->stream));
  FH_HIP(hipEventRecord(m1, ctx->stream));
  FH_HIP(hipStreamSynchronize(ctx->stream));
  float ms_m = 0.0f;
  FH_HIP(hipEventElapsedTime(&ms_m, m0, m1));
  (void)hipEventDestroy(m0); (void)hipEventDestroy(m1);
  const double memcpy_gbs = 2.0 * (double)n * 16.0 * iters / (ms_m * 1e-3) / 1e9;
  if (memcpy_gbs > *copy_gbs) *copy_gbs = memcpy_gbs;
  return FH_OK;
}
int fh_kat_tex2d(fh_ctx* ctx, const uint8_t* rgba8, const float* rgba32f, uint32_t width, uint32_t height, int srgb, uint32_t n, const float* uv2, float* out4)
{
  KCTX(ctx);
  if ((!rgba8 && !rgba32f) || !uv2 || !out4 || width == 0 || height == 0) return fail(ctx, FH_E_INVALID, "fh_kat_tex2d: bad argument");
  Tmp<uint8_t> t8;
  Tmp<float> t32, lut, uv, o;
  const size_t texels = (size_t)width * height * 4;
  if (rgba8) FH_HIP(t8.up(rgba8, texels)); else FH_HIP(t32.up(rgba32f, texels));
  float h_lut[256];
  for (int i = 0; i < 256; ++i) h_lut[i] = fht_srgb_to_linear((float)i * (1.0f / 255.0f));
  FH_HIP(lut.up(h_lut, 256)); FH_HIP(uv.up(uv2, 2ull * n)); FH_HIP(o.up(nullptr, 4ull * n));
  const fht_texture tex{rgba8 ? t8.p : nullptr, rgba8 ? nullptr : t32.p, width, height, srgb ? 1u : 0u};
  hipLaunchKernelGGL(k_tex2d, dim3(blocks(n)), dim3(256), 0, ctx->stream, tex, lut.p, n, uv.p, o.p);
  FH_HIP(hipStreamSynchronize(ctx->stream));
  FH_HIP(o.down(out4));
  return FH_OK;
}

}  // extern "C"
