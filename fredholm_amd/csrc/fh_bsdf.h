// fh_bsdf.h -- layered Standard-Surface BSDF for the shade kernels, specialised at compile time
// by the set of lobes a material class can ever enable.
//
// Behavioural source: fredholm/modules/bsdf.cu:11-345 (mixing, lobe order coat, metal, specular,
// transmission, sheen, diffuse transmission, diffuse), bxdf.cu:81-116,151-299,428-822 (lobes),
// lut.cu:957-1081 (albedo tables), sampling.cu:112-150 (7-bin lobe selection).
// Instead of the reference's 408-byte object holding seven lobe objects, a shade thread keeps one
// flat context (roughness alphas, conductor n/k, Oren-Nayar A/B, three albedos, the 8-entry CDF)
// and the template mask LOBES removes the code of lobes the class cannot reach; a cleared bit is
// a host-side guarantee that the lobe's weight and enabling condition are exactly zero/false for
// every material of the class, so skipping its terms (+0 contributions) is result-identical.
// NaN behaviour is part of the contract: back-facing hits on opaque materials make every lobe
// weight 0, the CDF NaN, the pdfs NaN and (through clamp) the light weights 1 -- as in the reference.
#pragma once
#include "fh_device.h"
#include "fh_sampler.h"

namespace fh {

enum : uint32_t { L_COAT = 1, L_METAL = 2, L_SPEC = 4, L_TRANS = 8, L_SHEEN = 16, L_DT = 32, L_DIFF = 64, L_ALL = 127 };

struct MatParams {  // the reference's ShadingParams (shared.h:173-199) after fill_shading_params
  float diffuse; f3 base_color; float diffuse_roughness;
  float specular; f3 specular_color; float specular_roughness;
  float metalness;
  float coat; f3 coat_color; float coat_roughness;
  float transmission; f3 transmission_color;
  float sheen; f3 sheen_color; float sheen_roughness;
  float subsurface; f3 subsurface_color;
  float thin_walled;
};


FH_HD float abs_cos(f3 w) { return fabsf(w.y); }
FH_HD float sin_t(f3 w) { return sqrt_cr(fmaxf(1.0f - w.y * w.y, 0.0f)); }
FH_HD float sin_p(f3 w) { return w.z / sqrt_cr(fmaxf(1.0f - w.y * w.y, 0.0f)); }
FH_HD float cos_p(f3 w) { return w.x / sqrt_cr(fmaxf(1.0f - w.y * w.y, 0.0f)); }
FH_HD f3 reflect_about(f3 w, f3 n) { return normalize(-w + 2.0f * dot(w, n) * n); }
FH_HD bool refract_through(f3 w, f3 n, float ni, float nt, f3& wt)
{
  const f3 th = -ni / nt * (w - dot(w, n) * n);
  if (dot(th, th) > 1.0f) return false;
  const f3 tp = -sqrt_cr(fmaxf(1.0f - dot(th, th), 0.0f)) * n;
  wt = th + tp;
  return true;
}
FH_HD float fresnel_dielectric(float c, float ior)
{
  const float temp = ior * ior + c * c - 1.0f;
  if (temp < 0.0f) return 1.0f;
  const float g = sqrt_cr(temp);
  const float t0 = (g - c) / (g + c);
  const float t1 = ((g + c) * c - 1.0f) / ((g - c) * c + 1.0f);
  return 0.5f * t0 * t0 * (1.0f + t1 * t1);
}
FH_HD f3 fresnel_conductor(float c, f3 ior, f3 k)
{
  const float c2 = c * c;
  const f3 two_eta_cos = 2.0f * ior * c;
  const f3 t0 = ior * ior + k * k;
  const f3 t1 = t0 * c2;
  const f3 Rs = (t0 - two_eta_cos + c2) / (t0 + two_eta_cos + c2);
  const f3 Rp = (t1 - two_eta_cos + 1.0f) / (t1 + two_eta_cos + 1.0f);
  return 0.5f * (Rp + Rs);
}
// GGX terms for an isotropic alpha pair (ax == ay numerically, kept separate to mirror the maths)
FH_HD float ggx_D(float ax, float ay, f3 wh)
{
  const float t = wh.x * wh.x / (ax * ax) + wh.z * wh.z / (ay * ay) + wh.y * wh.y;
  return (float)(1.0f / (3.14159265358979323846 * ax * ay * t * t));  // fp64, as the reference's M_PI forces
}
FH_HD float ggx_lambda(float ax, float ay, f3 w)
{
  const float t = (ax * ax * w.x * w.x + ay * ay * w.z * w.z) / (w.y * w.y);
  return 0.5f * (-1.0f + sqrt_cr(1.0f + t));
}
FH_HD float ggx_G2(float ax, float ay, f3 wo, f3 wi) { return 1.0f / (1.0f + ggx_lambda(ax, ay, wo) + ggx_lambda(ax, ay, wi)); }
FH_HD float ggx_Dvis(float ax, float ay, f3 w, f3 wh)
{
  const float g1 = 1.0f / (1.0f + ggx_lambda(ax, ay, w));
  return g1 * fabsf(dot(w, wh)) * ggx_D(ax, ay, wh) / abs_cos(w);
}
FH_HD f3 zero_if_bad(f3 v) { return bad3(v) ? mk3(0.0f) : v; }
FH_HD float zero_if_bad(float v) { return bad1(v) ? 0.0f : v; }

FH_HD f2 lut2_at(const float* t, int a, int b) { a = clampi(a, 0, 15); b = clampi(b, 0, 15); const int idx = 2 * a + 32 * b; return mk2(t[idx], t[idx + 1]); }
FH_HD float lut_reflection_albedo(const float* tbl, f3 w, float rough, float F0)
{
  const float u = fabsf(w.y), v = clampf(rough, 0.0f, 1.0f);
  const int i = clampi((int)(u * 16), 0, 15), j = clampi((int)(v * 16), 0, 15);
  const f2 t0 = lut2_at(tbl, i, j), t1 = lut2_at(tbl, i + 1, j), t2 = lut2_at(tbl, i, j + 1), t3 = lut2_at(tbl, i + 1, j + 1);
  const float hx = u * 16 - i, hy = v * 16 - j;
  const f2 tx0 = (1.0f - hx) * t0 + hx * t1;
  const f2 tx1 = (1.0f - hx) * t2 + hx * t3;
  const f2 rg = (1.0f - hy) * tx0 + hy * tx1;
  return F0 * rg.x + (1.0f - F0) * rg.y;
}
FH_HD float lut1_at(const float* t, int a, int b) { a = clampi(a, 0, 15); b = clampi(b, 0, 15); return t[a + 16 * b]; }
FH_HD float lut_sheen_albedo(const float* tbl, f3 w, float rough)
{
  const float u = fabsf(w.y), v = clampf(rough, 0.0f, 1.0f);
  const int i = clampi((int)(u * 16), 0, 15), j = clampi((int)(v * 16), 0, 15);
  const float t0 = lut1_at(tbl, i, j), t1 = lut1_at(tbl, i + 1, j), t2 = lut1_at(tbl, i, j + 1), t3 = lut1_at(tbl, i + 1, j + 1);
  const float hx = u * 16 - i, hy = v * 16 - j;
  const float tx0 = (1.0f - hx) * t0 + hx * t1;
  const float tx1 = (1.0f - hx) * t2 + hx * t3;
  return (1.0f - hy) * tx0 + hy * tx1;
}

template <uint32_t LOBES>
struct Bsdf {
  MatParams p;
  float ni, nt, eta;
  f3 coat_absorption;
  float coat_lum, spec_lum, sheen_lum, coat_albedo, spec_albedo, sheen_albedo;
  float cdf[8];
  float a_coat, a_spec;  // GGX alpha (x == y)
  f3 metal_n, metal_k;
  float on_A, on_B;

  static constexpr bool has(uint32_t l) { return (LOBES & l) != 0; }

  FH_HD void init(f3 wo, const MatParams& sp, bool entering, const BsdfTables& tb)
  {
    p = sp;
    ni = entering ? 1.0f : 1.5f;
    nt = entering ? 1.5f : 1.0f;
    eta = nt / ni;
    coat_lum = lum(p.coat_color);
    spec_lum = lum(p.specular_color);
    sheen_lum = lum(p.sheen_color);
    coat_albedo = spec_albedo = sheen_albedo = 0.0f;
    coat_absorption = mk3(1.0f) + p.coat * (p.coat_color * (1.0f - coat_albedo) - mk3(1.0f));
    const float tF = (nt - ni) / (nt + ni);
    const float F0 = tF * tF;
    if (has(L_COAT) && p.coat * coat_lum > 0.0f) coat_albedo = entering ? lut_reflection_albedo(tb.reflection, wo, p.coat_roughness, F0) : 0.0f;
    if (has(L_SPEC) && p.specular * spec_lum > 0.0f) spec_albedo = eta >= 1.0f ? lut_reflection_albedo(tb.reflection, wo, p.specular_roughness, F0) : 0.0f;
    if (has(L_SHEEN) && (p.sheen * sheen_lum) != 0.0f) sheen_albedo = entering ? lut_sheen_albedo(tb.sheen, wo, p.sheen_roughness) : 0.0f;
    p.coat = entering ? p.coat : 0.0f;
    p.metalness = entering ? p.metalness : 0.0f;
    p.specular = entering ? p.specular : 0.0f;
    p.sheen = entering ? p.sheen : 0.0f;
    p.diffuse = entering ? p.diffuse : 0.0f;
    float w[7];
    const float nc = 1.0f - p.coat * coat_albedo;
    w[0] = p.coat * coat_albedo;
    w[1] = nc * p.metalness;
    w[2] = nc * (1.0f - p.metalness) * p.specular * spec_albedo;
    w[3] = nc * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * p.transmission;
    w[4] = nc * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * p.sheen * sheen_albedo;
    w[5] = nc * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * (1.0f - p.transmission) * (1.0f - p.sheen * sheen_albedo) * p.subsurface * p.thin_walled;
    w[6] = nc * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * (1.0f - p.transmission) * (1.0f - p.sheen * sheen_albedo) * (1.0f - p.subsurface) * p.diffuse;
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) sum += w[i];
    cdf[0] = 0.0f;
#pragma unroll
    for (int i = 1; i < 8; ++i) cdf[i] = cdf[i - 1] + w[i - 1] / sum;
    a_coat = p.coat_roughness * p.coat_roughness * (1.0f + 0.0f);
    a_spec = p.specular_roughness * p.specular_roughness * (1.0f + 0.0f);
    if (has(L_METAL)) {
      const f3 refl = mk3(clampf(p.base_color.x, 0.0f, 0.99f), clampf(p.base_color.y, 0.0f, 0.99f), clampf(p.base_color.z, 0.0f, 0.99f));
      const f3 tint = mk3(clampf(p.specular_color.x, 0.0f, 0.99f), clampf(p.specular_color.y, 0.0f, 0.99f), clampf(p.specular_color.z, 0.0f, 0.99f));
      const f3 rs = sqrt3(refl);
      metal_n = tint * (1.0f - refl) / (1.0f + refl) + (1.0f - tint) * (1.0f + rs) / (1.0f - rs);
      const f3 t1 = metal_n + 1.0f;
      const f3 t2 = metal_n - 1.0f;
      metal_k = sqrt3((refl * (t1 * t1) - t2 * t2) / (1.0f - refl));
    }
    const float s2 = p.diffuse_roughness * p.diffuse_roughness;
    on_A = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
    on_B = 0.45f * s2 / (s2 + 0.09f);
  }

  FH_HD float pmf(int i) const { return cdf[i + 1] - cdf[i]; }

  // ---- lobes
  FH_HD f3 ggx_refl_dielectric_eval(float a, f3 wo, f3 wi) const
  {
    const f3 wh = normalize(wo + wi);
    const f3 f = mk3(fresnel_dielectric(fabsf(dot(wo, wh)), eta));
    const float d = ggx_D(a, a, wh), g = ggx_G2(a, a, wo, wi);
    return 0.25f * (f * d * g) / (abs_cos(wo) * abs_cos(wi));
  }
  FH_HD f3 metal_eval(f3 wo, f3 wi) const
  {
    const f3 wh = normalize(wo + wi);
    const f3 f = fresnel_conductor(fabsf(dot(wo, wh)), metal_n, metal_k);
    const float d = ggx_D(a_spec, a_spec, wh), g = ggx_G2(a_spec, a_spec, wo, wi);
    return 0.25f * (f * d * g) / (abs_cos(wo) * abs_cos(wi));
  }
  FH_HD float ggx_refl_pdf(float a, f3 wo, f3 wi) const
  {
    const f3 wh = normalize(wo + wi);
    return 0.25f * ggx_Dvis(a, a, wo, wh) / fabsf(dot(wo, wh));
  }
  FH_HD f3 trans_half(f3 wo, f3 wi) const
  {
    f3 wh = normalize(-(ni * wo + nt * wi));
    if (wh.y < 0.0f) wh = -wh;
    return wh;
  }
  FH_HD f3 trans_eval(f3 wo, f3 wi) const
  {
    const f3 wh = trans_half(wo, wi);
    const f3 f = mk3(fresnel_dielectric(fabsf(dot(wo, wh)), nt / ni));
    const float d = ggx_D(a_spec, a_spec, wh), g = ggx_G2(a_spec, a_spec, wo, wi);
    const float odh = dot(wo, wh), idh = dot(wi, wh);
    const float t = ni * odh + nt * idh;
    return fabsf(odh) * fabsf(idh) * nt * nt * max3(1.0f - f, mk3(0.0f)) * g * d / (abs_cos(wo) * abs_cos(wi) * t * t);
  }
  FH_HD float trans_pdf(f3 wo, f3 wi) const
  {
    const f3 wh = trans_half(wo, wi);
    const float idh = dot(wi, wh);
    const float t = ni * dot(wo, wh) + nt * idh;
    return ggx_Dvis(a_spec, a_spec, wo, wh) * nt * nt * fabsf(idh) / (t * t);
  }
  static FH_HD float sheen_interp(float r, float p0, float p1) { const float t = 1.0f - r; const float t2 = t * t; return t2 * p0 + (1.0f - t2) * p1; }
  FH_HD float sheen_L(float x) const
  {
    const float r = p.sheen_roughness;
    const float a = sheen_interp(r, 25.3245, 21.5473), b = sheen_interp(r, 3.32435, 3.82987), c = sheen_interp(r, 0.16801, 0.19823);
    const float d = sheen_interp(r, -1.27393, -1.97760), e = sheen_interp(r, -4.85967, -4.32054);
    return a / (1.0f + b * fhe_pow(x, c)) + d * x + e;
  }
  FH_HD float sheen_lambda(f3 w) const
  {
    const float c = abs_cos(w);
    return (c < 0.5f) ? fhe_exp(sheen_L(c)) : fhe_exp(2.0f * sheen_L(0.5f) - sheen_L(1.0f - c));
  }
  FH_HD f3 sheen_eval(f3 wo, f3 wi) const
  {
    const f3 wh = normalize(wo + wi);
    const float s = fabsf(sin_t(wh));
    const float d = (2.0f + 1.0f / p.sheen_roughness) * fhe_pow(s, 1.0f / p.sheen_roughness) / (2.0f * kPi);
    const float g = 1.0f / (1.0f + sheen_lambda(wo) + sheen_lambda(wi));
    const float f = 1.0f;
    return mk3(0.25f * (f * d * g) / (abs_cos(wo) * abs_cos(wi)));
  }
  FH_HD f3 oren_nayar_eval(f3 wo, f3 wi) const
  {
    const float sto = sin_t(wo), sti = sin_t(wi);
    float cmax = 0.0f;
    if (sti > 1e-4f && sto > 1e-4f) {
      const float spo = sin_p(wo), cpo = cos_p(wo);
      const float spi = sin_p(wi), cpi = cos_p(wi);
      cmax = fmaxf(cpi * cpo + spi * spo, 0.0f);
    }
    const bool b = abs_cos(wi) > abs_cos(wo);
    const float s_alpha = b ? sto : sti;
    const float t_beta = b ? sti / abs_cos(wi) : sto / abs_cos(wo);
    return p.base_color * (on_A + on_B * cmax * s_alpha * t_beta) / kPi;
  }

  // ---- mixture
  FH_HD f3 eval(f3 wo, f3 wi) const
  {
    f3 coat = mk3(0.0f), metal = mk3(0.0f), spec = mk3(0.0f), trans = mk3(0.0f), sheen = mk3(0.0f), dt = mk3(0.0f), dr = mk3(0.0f);
    if (has(L_COAT) && p.coat * coat_lum > 0.0f) coat = zero_if_bad(ggx_refl_dielectric_eval(a_coat, wo, wi));
    if (has(L_METAL) && p.metalness > 0.0f) metal = zero_if_bad(metal_eval(wo, wi));
    if (has(L_SPEC) && p.specular * spec_lum > 0.0f) spec = zero_if_bad(ggx_refl_dielectric_eval(a_spec, wo, wi));
    if (has(L_TRANS) && p.transmission > 0.0f) trans = zero_if_bad(trans_eval(wo, wi));
    if (has(L_SHEEN) && p.sheen * sheen_lum > 0.0f) sheen = zero_if_bad(sheen_eval(wo, wi));
    if (has(L_DT) && p.subsurface * p.thin_walled > 0.0f) dt = zero_if_bad(oren_nayar_eval(wo, wi));
    if (has(L_DIFF) && p.diffuse > 0.0f) dr = zero_if_bad(oren_nayar_eval(wo, wi));
    f3 ret = mk3(0.0f), m = mk3(1.0f);
    ret += p.coat * coat;
    m *= coat_absorption;
    ret += m * p.metalness * metal;
    m *= mk3(1.0f - p.metalness);
    ret += m * p.specular * p.specular_color * spec;
    m *= (1.0f - p.specular * p.specular_color * spec_albedo);
    ret += m * p.transmission * p.transmission_color * trans;
    m *= mk3(1.0f - p.transmission);
    ret += m * p.sheen * p.sheen_color * sheen;
    m *= mk3(1.0f - p.sheen * sheen_albedo);
    ret += m * p.subsurface * p.subsurface_color * p.thin_walled * dt;
    m *= mk3(1.0f - p.subsurface);
    ret += m * p.diffuse * dr;
    return ret;
  }

  FH_HD float eval_pdf(f3 wo, f3 wi) const
  {
    float coat = 0, metal = 0, spec = 0, trans = 0, sheen = 0, dt = 0, dr = 0;
    if (has(L_COAT) && p.coat * coat_lum > 0.0f) coat = zero_if_bad(ggx_refl_pdf(a_coat, wo, wi));
    if (has(L_METAL) && p.metalness > 0.0f) metal = zero_if_bad(ggx_refl_pdf(a_spec, wo, wi));
    if (has(L_SPEC) && p.specular * spec_lum > 0.0f) spec = zero_if_bad(ggx_refl_pdf(a_spec, wo, wi));
    if (has(L_TRANS) && p.transmission > 0.0f) trans = zero_if_bad(trans_pdf(wo, wi));
    if (has(L_SHEEN) && p.sheen * sheen_lum > 0.0f) sheen = zero_if_bad(abs_cos(wi) / kPi);
    if (has(L_DT) && p.subsurface * p.thin_walled > 0.0f) dt = zero_if_bad(abs_cos(wi) / kPi);
    if (has(L_DIFF) && p.diffuse > 0.0f) dr = zero_if_bad(abs_cos(wi) / kPi);
    return pmf(0) * coat + pmf(1) * metal + pmf(2) * spec + pmf(3) * trans + pmf(4) * sheen + pmf(5) * dt + pmf(6) * dr;
  }

  FH_HD f3 sample(f3 wo, float u, f2 v, f3& f, float& pdf) const
  {
    int idx = 6;
    float pm = cdf[7] - cdf[6];
    {
      float c = 0.0f;
      bool found = false;
#pragma unroll
      for (int i = 1; i <= 7; ++i) {
        c += cdf[i] - cdf[i - 1];
        if (!found && u < c) { pm = cdf[i] - cdf[i - 1]; idx = i - 1; found = true; }
      }
    }
    f3 wi;
    const f3 pre = coat_absorption * (1.0f - p.metalness);  // common prefix of lobes 2..6
    if (has(L_COAT) && idx == 0) {
      const f3 wh = sample_vndf(wo, a_coat, a_coat, v);
      wi = reflect_about(wo, wh);
      f = ggx_refl_dielectric_eval(a_coat, wo, wi);
      pdf = ggx_refl_pdf(a_coat, wo, wi);
      f = f * p.coat;
    } else if (has(L_METAL) && idx == 1) {
      const f3 wh = sample_vndf(wo, a_spec, a_spec, v);
      wi = reflect_about(wo, wh);
      f = metal_eval(wo, wi);
      pdf = ggx_refl_pdf(a_spec, wo, wi);
      f *= coat_absorption * p.metalness;
    } else if (has(L_SPEC) && idx == 2) {
      const f3 wh = sample_vndf(wo, a_spec, a_spec, v);
      wi = reflect_about(wo, wh);
      f = ggx_refl_dielectric_eval(a_spec, wo, wi);
      pdf = ggx_refl_pdf(a_spec, wo, wi);
      f *= pre * p.specular * p.specular_color;
    } else if (has(L_TRANS) && idx == 3) {
      const f3 wh = sample_vndf(wo, a_spec, a_spec, v);
      if (!refract_through(wo, wh, ni, nt, wi)) {
        wi = reflect_about(wo, wh);
        const f3 fr = mk3(fresnel_dielectric(fabsf(dot(wo, wh)), nt / ni));
        const float d = ggx_D(a_spec, a_spec, wh), g = ggx_G2(a_spec, a_spec, wo, wi);
        f = 0.25f * (fr * d * g) / (abs_cos(wo) * abs_cos(wi));
        pdf = 0.25f * ggx_Dvis(a_spec, a_spec, wo, wh) / fabsf(dot(wi, wh));
      } else {
        f = trans_eval(wo, wi);
        pdf = trans_pdf(wo, wi);
      }
      f *= pre * (1.0f - p.specular * p.specular_color * spec_albedo) * p.transmission * p.transmission_color;
    } else if (has(L_SHEEN) && idx == 4) {
      const f3 wh = cosine_hemisphere(v);
      wi = reflect_about(wo, wh);
      f = sheen_eval(wo, wi);
      pdf = abs_cos(wi) / kPi;
      f *= pre * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) * p.sheen * p.sheen_color;
    } else if (has(L_DT) && idx == 5) {
      wi = -cosine_hemisphere(v);
      f = oren_nayar_eval(wo, wi);
      pdf = abs_cos(wi) / kPi;
      f *= pre * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) * (1.0f - p.sheen * sheen_albedo) * p.subsurface * p.subsurface_color * p.thin_walled;
    } else {
      // lobe 6 (diffuse reflection); also the reference's fall-through when the CDF is NaN
      wi = cosine_hemisphere(v);
      f = oren_nayar_eval(wo, wi);
      pdf = abs_cos(wi) / kPi;
      f *= pre * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) * (1.0f - p.sheen * sheen_albedo) * (1.0f - p.subsurface) * p.diffuse;
    }
    pdf *= pm;
    return wi;
  }
};

}  // namespace fh
