// oracle/obsdf.h -- TEST INFRASTRUCTURE (CPU checker), never linked into the product.
//
// Restates the reference's Standard-Surface-like layered BSDF:
//   lobes   fredholm/modules/bxdf.cu:9-116 (trig helpers, reflect/refract, metallic Fresnel),
//           :151-205 Oren-Nayar, :209-264 diffuse transmission, :274-299 Fresnel terms,
//           :428-518 GGX dielectric reflection, :522-611 GGX conductor, :615-740 Walter
//           transmission, :743-822 Estevez-Kulla sheen
//   LUTs    fredholm/modules/lut.cu:957-1081 (bilinear fetch of 16x16 tables)
//   mixing  fredholm/modules/bsdf.cu:11-127 (ctor), :129-212 eval, :214-293 sample, :295-345 pdf
// Thin-film (fresnel_airy, bxdf.cu:301-424) is never enabled by the reference's ctor calls
// (thickness defaults to 0) and is omitted.  Quirks kept: coat absorption uses the
// not-yet-computed coat albedo (bsdf.cu:27-30), D() is evaluated in double because of M_PI
// (bxdf.cu:488), an all-zero lobe table yields NaN pmfs (sampling.cu:116-128).
#pragma once
#include "osampler.h"

namespace orc {

struct ShadingParams {  // shared.h:173-199
  float diffuse = 1.0f; V3 base_color = {0, 0, 0}; float diffuse_roughness = 0.0f;
  float specular = 1.0f; V3 specular_color = {0, 0, 0}; float specular_roughness = 0.2f;
  float metalness = 0.0f;
  float coat = 0.0f; V3 coat_color = {1, 1, 1}; float coat_roughness = 0.1f;
  float transmission = 0; V3 transmission_color = {1, 1, 1};
  float sheen = 0.0f; V3 sheen_color = {1, 1, 1}; float sheen_roughness = 0.3f;
  float subsurface = 0; V3 subsurface_color = {1, 1, 1};
  float thin_walled = 0.0f;
};

extern const float* g_lut_reflection;  // 16*16*2
extern const float* g_lut_sheen;       // 16*16

inline float luminance(V3 c) { return dot(c, v3(0.2126729f, 0.7151522f, 0.0721750f)); }  // math.cu:90-93

inline float abs_cos(V3 w) { return fabsf(w.y); }
inline float sin_t(V3 w) { return sqrtf(fmaxf(1.0f - w.y * w.y, 0.0f)); }
inline float sin_p(V3 w) { return w.z / sqrtf(fmaxf(1.0f - w.y * w.y, 0.0f)); }
inline float cos_p(V3 w) { return w.x / sqrtf(fmaxf(1.0f - w.y * w.y, 0.0f)); }

inline V3 reflect_about(V3 w, V3 n) { return normalize(-w + 2.0f * dot(w, n) * n); }  // bxdf.cu:81-84
inline bool refract_through(V3 w, V3 n, float ni, float nt, V3& wt)               // bxdf.cu:86-94
{
  const V3 th = -ni / nt * (w - dot(w, n) * n);
  if (dot(th, th) > 1.0f) return false;
  const V3 tp = -sqrtf(fmaxf(1.0f - dot(th, th), 0.0f)) * n;
  wt = th + tp;
  return true;
}
inline V2 roughness_to_alpha(float r, float aniso) { return v2(r * r * (1.0f + aniso), r * r * (1.0f - aniso)); }  // :96-104

// bxdf.cu:107-116
inline void metallic_fresnel(V3 refl, V3 tint, V3& n, V3& k)
{
  const V3 rs = sqrt3(refl);
  n = tint * (1.0f - refl) / (1.0f + refl) + (1.0f - tint) * (1.0f + rs) / (1.0f - rs);
  const V3 t1 = n + 1.0f;
  const V3 t2 = n - 1.0f;
  k = sqrt3((refl * (t1 * t1) - t2 * t2) / (1.0f - refl));
}

// bxdf.cu:274-283
inline float fresnel_dielectric(float c, float ior)
{
  const float temp = ior * ior + c * c - 1.0f;
  if (temp < 0.0f) return 1.0f;
  const float g = sqrtf(temp);
  const float t0 = (g - c) / (g + c);
  const float t1 = ((g + c) * c - 1.0f) / ((g - c) * c + 1.0f);
  return 0.5f * t0 * t0 * (1.0f + t1 * t1);
}
// bxdf.cu:286-299
inline V3 fresnel_conductor(float c, V3 ior, V3 k)
{
  const float c2 = c * c;
  const V3 two_eta_cos = 2.0f * ior * c;
  const V3 t0 = ior * ior + k * k;
  const V3 t1 = t0 * c2;
  const V3 Rs = (t0 - two_eta_cos + c2) / (t0 + two_eta_cos + c2);
  const V3 Rp = (t1 - two_eta_cos + 1.0f) / (t1 + two_eta_cos + 1.0f);
  return 0.5f * (Rp + Rs);
}

// GGX pieces shared by the three microfacet lobes (bxdf.cu:484-512 and copies)
struct Ggx {
  V2 a;
  float D(V3 wh) const
  {
    const float t = wh.x * wh.x / (a.x * a.x) + wh.z * wh.z / (a.y * a.y) + wh.y * wh.y;
    return (float)(1.0f / (M_PI * a.x * a.y * t * t));  // double arithmetic, as in the reference
  }
  float lambda(V3 w) const
  {
    const float t = (a.x * a.x * w.x * w.x + a.y * a.y * w.z * w.z) / (w.y * w.y);
    return 0.5f * (-1.0f + sqrtf(1.0f + t));
  }
  float G1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
  float G2(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
  float Dvis(V3 w, V3 wh) const { return G1(w) * fabsf(dot(w, wh)) * D(wh) / abs_cos(w); }
};

// Oren-Nayar (bxdf.cu:151-205) and its flipped transmission twin (:209-264)
struct OrenNayar {
  V3 albedo; float A, B;
  void init(V3 alb, float rough)
  {
    albedo = alb;
    const float s2 = rough * rough;
    A = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
    B = 0.45f * s2 / (s2 + 0.09f);
  }
  V3 eval(V3 wo, V3 wi) const
  {
    const float sto = sin_t(wo), sti = sin_t(wi);
    float cmax = 0.0f;
    if (sti > 1e-4f && sto > 1e-4f) {
      const float spo = sin_p(wo), cpo = cos_p(wo);
      const float spi = sin_p(wi), cpi = cos_p(wi);
      const float c = cpi * cpo + spi * spo;
      cmax = fmaxf(c, 0.0f);
    }
    const bool b = abs_cos(wi) > abs_cos(wo);
    const float s_alpha = b ? sto : sti;
    const float t_beta = b ? sti / abs_cos(wi) : sto / abs_cos(wo);
    return albedo * (A + B * cmax * s_alpha * t_beta) / kPi;
  }
  V3 sample(V3 wo, V2 u, V3& f, float& pdf, bool flip) const
  {
    V3 wi = cosine_hemisphere(u);
    if (flip) wi = -wi;
    f = eval(wo, wi);
    pdf = abs_cos(wi) / kPi;
    return wi;
  }
  float pdf(V3, V3 wi) const { return abs_cos(wi) / kPi; }
};

// GGX reflection with dielectric Fresnel (bxdf.cu:428-518)
struct GgxDielectric {
  float ior; Ggx g;
  void init(float ior_, float rough) { ior = ior_; g.a = roughness_to_alpha(rough, 0.0f); }
  V3 eval(V3 wo, V3 wi) const
  {
    const V3 wh = normalize(wo + wi);
    const V3 f = v3(fresnel_dielectric(fabsf(dot(wo, wh)), ior));
    const float d = g.D(wh), gg = g.G2(wo, wi);
    return 0.25f * (f * d * gg) / (abs_cos(wo) * abs_cos(wi));
  }
  float pdf(V3 wo, V3 wi) const
  {
    const V3 wh = normalize(wo + wi);
    return 0.25f * g.Dvis(wo, wh) / fabsf(dot(wo, wh));
  }
  V3 sample(V3 wo, V2 u, V3& f, float& p) const
  {
    const V3 wh = vndf(wo, g.a, u);
    const V3 wi = reflect_about(wo, wh);
    f = eval(wo, wi);
    p = pdf(wo, wi);
    return wi;
  }
};

// GGX reflection with conductor Fresnel (bxdf.cu:522-611)
struct GgxConductor {
  V3 n, k; Ggx g;
  void init(V3 n_, V3 k_, float rough) { n = n_; k = k_; g.a = roughness_to_alpha(rough, 0.0f); }
  V3 eval(V3 wo, V3 wi) const
  {
    const V3 wh = normalize(wo + wi);
    const V3 f = fresnel_conductor(fabsf(dot(wo, wh)), n, k);
    const float d = g.D(wh), gg = g.G2(wo, wi);
    return 0.25f * (f * d * gg) / (abs_cos(wo) * abs_cos(wi));
  }
  float pdf(V3 wo, V3 wi) const
  {
    const V3 wh = normalize(wo + wi);
    return 0.25f * g.Dvis(wo, wh) / fabsf(dot(wo, wh));
  }
  V3 sample(V3 wo, V2 u, V3& f, float& p) const
  {
    const V3 wh = vndf(wo, g.a, u);
    const V3 wi = reflect_about(wo, wh);
    f = eval(wo, wi);
    p = pdf(wo, wi);
    return wi;
  }
};

// Walter et al. rough transmission (bxdf.cu:615-740)
struct GgxTransmission {
  float ni, nt; Ggx g;
  void init(float ni_, float nt_, float rough) { ni = ni_; nt = nt_; g.a = roughness_to_alpha(rough, 0.0f); }
  V3 half_vector(V3 wo, V3 wi) const
  {
    V3 wh = normalize(-(ni * wo + nt * wi));
    if (wh.y < 0.0f) wh = -wh;
    return wh;
  }
  V3 eval(V3 wo, V3 wi) const
  {
    const V3 wh = half_vector(wo, wi);
    const V3 f = v3(fresnel_dielectric(fabsf(dot(wo, wh)), nt / ni));
    const float d = g.D(wh), gg = g.G2(wo, wi);
    const float odh = dot(wo, wh), idh = dot(wi, wh);
    const float t = ni * odh + nt * idh;
    return fabsf(odh) * fabsf(idh) * nt * nt * fmax3(1.0f - f, v3(0.0f)) * gg * d / (abs_cos(wo) * abs_cos(wi) * t * t);
  }
  float pdf(V3 wo, V3 wi) const
  {
    const V3 wh = half_vector(wo, wi);
    const float idh = dot(wi, wh);
    const float t = ni * dot(wo, wh) + nt * idh;
    return g.Dvis(wo, wh) * nt * nt * fabsf(idh) / (t * t);
  }
  V3 sample(V3 wo, V2 u, V3& f, float& p) const
  {
    const V3 wh = vndf(wo, g.a, u);
    V3 wi;
    if (!refract_through(wo, wh, ni, nt, wi)) {  // total internal reflection, :660-679
      wi = reflect_about(wo, wh);
      const V3 fr = v3(fresnel_dielectric(fabsf(dot(wo, wh)), nt / ni));
      const float d = g.D(wh), gg = g.G2(wo, wi);
      f = 0.25f * (fr * d * gg) / (abs_cos(wo) * abs_cos(wi));
      p = 0.25f * g.Dvis(wo, wh) / fabsf(dot(wi, wh));
      return wi;
    }
    f = eval(wo, wi);
    p = pdf(wo, wi);
    return wi;
  }
};

// Estevez-Kulla sheen (bxdf.cu:743-822)
struct Sheen {
  float rough;
  static float interp(float r, float p0, float p1) { const float t = 1.0f - r; const float t2 = t * t; return t2 * p0 + (1.0f - t2) * p1; }
  float L(float x) const
  {
    const float a = interp(rough, 25.3245, 21.5473);
    const float b = interp(rough, 3.32435, 3.82987);
    const float c = interp(rough, 0.16801, 0.19823);
    const float d = interp(rough, -1.27393, -1.97760);
    const float e = interp(rough, -4.85967, -4.32054);
    return a / (1.0f + b * oe::pow(x, c)) + d * x + e;
  }
  float lambda(V3 w) const
  {
    const float c = abs_cos(w);
    return (c < 0.5f) ? oe::exp(L(c)) : oe::exp(2.0f * L(0.5f) - L(1.0f - c));
  }
  float D(V3 wh) const
  {
    const float s = fabsf(sin_t(wh));
    return (2.0f + 1.0f / rough) * oe::pow(s, 1.0f / rough) / (2.0f * kPi);
  }
  V3 eval(V3 wo, V3 wi) const
  {
    const V3 wh = normalize(wo + wi);
    const float f = 1.0f;
    const float d = D(wh);
    const float gg = 1.0f / (1.0f + lambda(wo) + lambda(wi));
    return v3(0.25f * (f * d * gg) / (abs_cos(wo) * abs_cos(wi)));
  }
  float pdf(V3, V3 wi) const { return abs_cos(wi) / kPi; }
  V3 sample(V3 wo, V2 u, V3& f, float& p) const
  {
    const V3 wh = cosine_hemisphere(u);
    const V3 wi = reflect_about(wo, wh);
    f = eval(wo, wi);
    p = pdf(wo, wi);
    return wi;
  }
};

// lut.cu:957-994 / :1047-1081
inline float lut_reflection_albedo(V3 w, float rough, float F0)
{
  const float u = fabsf(w.y), v = clampf(rough, 0.0f, 1.0f);
  const int i = clampi((int)(u * 16), 0, 15), j = clampi((int)(v * 16), 0, 15);
  auto at = [](int a, int b) { a = clampi(a, 0, 15); b = clampi(b, 0, 15); const int idx = 2 * a + 32 * b; return v2(g_lut_reflection[idx], g_lut_reflection[idx + 1]); };
  const V2 t0 = at(i, j), t1 = at(i + 1, j), t2 = at(i, j + 1), t3 = at(i + 1, j + 1);
  const float hx = u * 16 - i, hy = v * 16 - j;
  const V2 tx0 = (1.0f - hx) * t0 + hx * t1;
  const V2 tx1 = (1.0f - hx) * t2 + hx * t3;
  const V2 rg = (1.0f - hy) * tx0 + hy * tx1;
  return F0 * rg.x + (1.0f - F0) * rg.y;
}
inline float lut_sheen_albedo(V3 w, float rough)
{
  const float u = fabsf(w.y), v = clampf(rough, 0.0f, 1.0f);
  const int i = clampi((int)(u * 16), 0, 15), j = clampi((int)(v * 16), 0, 15);
  auto at = [](int a, int b) { a = clampi(a, 0, 15); b = clampi(b, 0, 15); return g_lut_sheen[a + 16 * b]; };
  const float t0 = at(i, j), t1 = at(i + 1, j), t2 = at(i, j + 1), t3 = at(i + 1, j + 1);
  const float hx = u * 16 - i, hy = v * 16 - j;
  const float tx0 = (1.0f - hx) * t0 + hx * t1;
  const float tx1 = (1.0f - hx) * t2 + hx * t3;
  return (1.0f - hy) * tx0 + hy * tx1;
}

inline V3 zero_if_bad(V3 v) { return (anyinf(v) || anynan(v)) ? v3(0.0f) : v; }
inline float zero_if_bad(float v) { return (std::isinf(v) || std::isnan(v)) ? 0.0f : v; }

// bsdf.cu:8-379
struct Bsdf {
  ShadingParams p;
  float ni, nt, eta;
  GgxDielectric coat_l, spec_l; GgxConductor metal_l; GgxTransmission trans_l; Sheen sheen_l; OrenNayar dt_l, diff_l;
  V3 coat_absorption = {1, 1, 1};
  float coat_lum = 0, coat_albedo = 0, spec_lum = 0, spec_albedo = 0, sheen_lum = 0, sheen_albedo = 0;
  Discrete7 dist;
  float weights[7];

  Bsdf(V3 wo, const ShadingParams& sp, bool entering) : p(sp)
  {
    ni = entering ? 1.0f : 1.5f;
    nt = entering ? 1.5f : 1.0f;
    eta = nt / ni;
    coat_lum = luminance(p.coat_color);
    spec_lum = luminance(p.specular_color);
    sheen_lum = luminance(p.sheen_color);
    coat_absorption = lerp3(v3(1.0f), p.coat_color * (1.0f - coat_albedo), p.coat);  // albedo still 0 here
    const float tF = (nt - ni) / (nt + ni);
    const float F0 = tF * tF;
    if (p.coat * coat_lum > 0.0f) coat_albedo = entering ? lut_reflection_albedo(wo, p.coat_roughness, F0) : 0.0f;
    if (p.specular * spec_lum > 0.0f) spec_albedo = eta >= 1.0f ? lut_reflection_albedo(wo, p.specular_roughness, F0) : 0.0f;
    if ((p.sheen * sheen_lum) != 0.0f) sheen_albedo = entering ? lut_sheen_albedo(wo, p.sheen_roughness) : 0.0f;
    p.coat = entering ? p.coat : 0.0f;
    p.metalness = entering ? p.metalness : 0.0f;
    p.specular = entering ? p.specular : 0.0f;
    p.sheen = entering ? p.sheen : 0.0f;
    p.diffuse = entering ? p.diffuse : 0.0f;
    float* w = weights;
    w[0] = p.coat * coat_albedo;
    w[1] = (1.0f - p.coat * coat_albedo) * p.metalness;
    w[2] = (1.0f - p.coat * coat_albedo) * (1.0f - p.metalness) * p.specular * spec_albedo;
    w[3] = (1.0f - p.coat * coat_albedo) * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * p.transmission;
    w[4] = (1.0f - p.coat * coat_albedo) * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * p.sheen * sheen_albedo;
    w[5] = (1.0f - p.coat * coat_albedo) * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * (1.0f - p.transmission) *
           (1.0f - p.sheen * sheen_albedo) * p.subsurface * p.thin_walled;
    w[6] = (1.0f - p.coat * coat_albedo) * (1.0f - p.metalness) * (1.0f - p.specular * spec_albedo) * (1.0f - p.transmission) *
           (1.0f - p.sheen * sheen_albedo) * (1.0f - p.subsurface) * p.diffuse;
    dist.init(w);
    coat_l.init(eta, p.coat_roughness);
    spec_l.init(eta, p.specular_roughness);
    V3 n, k;
    metallic_fresnel(clamp3(p.base_color, v3(0), v3(0.99)), clamp3(p.specular_color, v3(0), v3(0.99)), n, k);
    metal_l.init(n, k, p.specular_roughness);
    trans_l.init(ni, nt, p.specular_roughness);
    sheen_l.rough = p.sheen_roughness;
    dt_l.init(p.base_color, p.diffuse_roughness);
    diff_l.init(p.base_color, p.diffuse_roughness);
  }

  V3 eval(V3 wo, V3 wi) const
  {
    V3 coat = v3(0.0f), metal = v3(0.0f), spec = v3(0.0f), trans = v3(0.0f), sheen = v3(0.0f), dt = v3(0.0f), dr = v3(0.0f);
    if (p.coat * coat_lum > 0.0f) coat = zero_if_bad(coat_l.eval(wo, wi));
    if (p.metalness > 0.0f) metal = zero_if_bad(metal_l.eval(wo, wi));
    if (p.specular * spec_lum > 0.0f) spec = zero_if_bad(spec_l.eval(wo, wi));
    if (p.transmission > 0.0f) trans = zero_if_bad(trans_l.eval(wo, wi));
    if (p.sheen * sheen_lum > 0.0f) sheen = zero_if_bad(sheen_l.eval(wo, wi));
    if (p.subsurface * p.thin_walled > 0.0f) dt = zero_if_bad(dt_l.eval(wo, wi));
    if (p.diffuse > 0.0f) dr = zero_if_bad(diff_l.eval(wo, wi));
    V3 ret = v3(0.0f), m = v3(1.0f);
    ret += p.coat * coat;
    m *= coat_absorption;
    ret += m * p.metalness * metal;
    m *= (1.0f - p.metalness);
    ret += m * p.specular * p.specular_color * spec;
    m *= (1.0f - p.specular * p.specular_color * spec_albedo);
    ret += m * p.transmission * p.transmission_color * trans;
    m *= (1.0f - p.transmission);
    ret += m * p.sheen * p.sheen_color * sheen;
    m *= (1.0f - p.sheen * sheen_albedo);
    ret += m * p.subsurface * p.subsurface_color * p.thin_walled * dt;
    m *= (1.0f - p.subsurface);
    ret += m * p.diffuse * dr;
    return ret;
  }

  V3 sample(V3 wo, float u, V2 v, V3& f, float& pdf) const
  {
    float pm;
    const int idx = dist.sample(u, pm);
    V3 wi;
    switch (idx) {
      case 0: wi = coat_l.sample(wo, v, f, pdf); f *= p.coat; break;
      case 1: wi = metal_l.sample(wo, v, f, pdf); f *= coat_absorption * p.metalness; break;
      case 2: wi = spec_l.sample(wo, v, f, pdf); f *= coat_absorption * (1.0f - p.metalness) * p.specular * p.specular_color; break;
      case 3:
        wi = trans_l.sample(wo, v, f, pdf);
        f *= coat_absorption * (1.0f - p.metalness) * (1.0f - p.specular * p.specular_color * spec_albedo) * p.transmission * p.transmission_color;
        break;
      case 4:
        wi = sheen_l.sample(wo, v, f, pdf);
        f *= coat_absorption * (1.0f - p.metalness) * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) * p.sheen * p.sheen_color;
        break;
      case 5:
        wi = dt_l.sample(wo, v, f, pdf, true);
        f *= coat_absorption * (1.0f - p.metalness) * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) *
             (1.0f - p.sheen * sheen_albedo) * p.subsurface * p.subsurface_color * p.thin_walled;
        break;
      default:
        wi = diff_l.sample(wo, v, f, pdf, false);
        f *= coat_absorption * (1.0f - p.metalness) * (1.0f - p.specular * p.specular_color * spec_albedo) * (1.0f - p.transmission) *
             (1.0f - p.sheen * sheen_albedo) * (1.0f - p.subsurface) * p.diffuse;
        break;
    }
    pdf *= pm;
    return wi;
  }

  float eval_pdf(V3 wo, V3 wi) const
  {
    float coat = 0, metal = 0, spec = 0, trans = 0, sheen = 0, dt = 0, dr = 0;
    if (p.coat * coat_lum > 0.0f) coat = zero_if_bad(coat_l.pdf(wo, wi));
    if (p.metalness > 0.0f) metal = zero_if_bad(metal_l.pdf(wo, wi));
    if (p.specular * spec_lum > 0.0f) spec = zero_if_bad(spec_l.pdf(wo, wi));
    if (p.transmission > 0.0f) trans = zero_if_bad(trans_l.pdf(wo, wi));
    if (p.sheen * sheen_lum > 0.0f) sheen = zero_if_bad(sheen_l.pdf(wo, wi));
    if (p.subsurface * p.thin_walled > 0.0f) dt = zero_if_bad(dt_l.pdf(wo, wi));
    if (p.diffuse > 0.0f) dr = zero_if_bad(diff_l.pdf(wo, wi));
    return dist.pmf(0) * coat + dist.pmf(1) * metal + dist.pmf(2) * spec + dist.pmf(3) * trans + dist.pmf(4) * sheen + dist.pmf(5) * dt + dist.pmf(6) * dr;
  }
};

}  // namespace orc
