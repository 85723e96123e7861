/* fh_elementary.h -- normative fp32 elementary functions (sin, cos, exp, log, pow, acos, atan2).
 *
 * The reference integrator calls CUDA libm (sinf/cosf/expf/powf/acosf/atan2f) on the device
 * (e.g. fredholm/modules/sampling.cu:54-64, arhosek.cu:103-118, bxdf.cu:784,791).  CUDA's
 * roundings cannot be reproduced without CUDA, and glibc's differ from ROCm's ocml, so a CPU
 * checker and a GPU kernel calling their own libm would disagree in the last bits -- and a path
 * tracer amplifies last-bit differences into different hit triangles.  This header fixes ONE
 * definition built only from IEEE-754 +,-,*,/,sqrt,fma and integer bit operations, which round
 * identically on x86-64 and gfx950 (HIP's default fp32 divide/sqrt are correctly rounded and
 * both sides are compiled with -ffp-contract=off).  Accuracy is <= 2 ulp over the argument
 * ranges the integrator uses (checked against float64 libm in tests/test_oracle_anchors.py:
 * test_elementary_accuracy).
 *
 * It is the PRODUCT's implementation of that numerical specification (host + device).  The CPU
 * checker has its own (oracle/oelementary.h: coefficient tables, one Horner routine, its own
 * special-case handling) and includes nothing from here; tests/test_oracle_anchors.py compiles this
 * header for the host and requires the two to agree bit for bit over millions of arguments, and
 * fh_kat_elementary does the same for the device build -- which is what lets images be compared
 * pixel by pixel instead of statistically.
 */
#ifndef FH_ELEMENTARY_H
#define FH_ELEMENTARY_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FHE_FN __host__ __device__ inline
#else
#define FHE_FN static inline
#endif

FHE_FN uint32_t fhe_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FHE_FN float fhe_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
FHE_FN uint64_t fhe_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
FHE_FN double fhe_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* sqrt(x), correctly rounded.  On the host: sqrtf.  On gfx950 the compiler's IEEE sqrtf is a ~20-instruction sequence; for 2^-100 <= x < inf and
 * for +-0 one Newton step on x * rsq(x) carried out with fused multiply-adds (Markstein's correction) gives the SAME bits in 9 -- compared over all 2^32
 * inputs on the device (tools/micro/sqrt_exhaustive.hip: 0 mismatches; below 2^-102 the residual would be denormal, hence the bound).  Every other input
 * (tiny, negative, infinite, NaN) takes sqrtf. */
FHE_FN float fhe_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_expect((__float_as_uint(x) - 0x0d800000u) < 0x72000000u || x == 0.0f, 1)) {
    const float y = __builtin_amdgcn_rsqf(fmaxf(x, 7.8886090522101181e-31f)); /* 2^-100: keeps +-0 away from rsq(0) = inf; s = +-0 * y stays +-0 */
    const float s = x * y, h = 0.5f * y;
    return fmaf(fmaf(-s, s, x), h, s);
  }
#endif
  return sqrtf(x);
}

/* round to nearest even, valid for |x| < 2^22 (magic-number trick: exact in IEEE fp32) */
FHE_FN float fhe_rint(float x)
{
  const float magic = 12582912.0f; /* 1.5 * 2^23 */
  return (x + magic) - magic;
}

/* sin and cos of x (radians), |x| <= ~1e4.  Cody-Waite 3-term reduction + cephes minimax kernels */
FHE_FN void fhe_sincos(float x, float* s, float* c)
{
  const float k = fhe_rint(x * 0.636619772367581343f); /* x * 2/pi */
  float r = fmaf(-k, 1.5703125f, x);                   /* pi/2 split into 3 parts */
  r = fmaf(-k, 4.83751296997070312e-4f, r);
  r = fmaf(-k, 7.54978995489188216e-8f, r);
  const float z = r * r;
  /* sin(r), |r| <= pi/4 */
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  const float sr = fmaf(z * r, ps, r);
  /* cos(r) */
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  const float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
  const int q = ((int)k) & 3;
  const float s0 = (q & 1) ? cr : sr;
  const float c0 = (q & 1) ? sr : cr;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
}
FHE_FN float fhe_sin(float x) { float s, c; fhe_sincos(x, &s, &c); return s; }
FHE_FN float fhe_cos(float x) { float s, c; fhe_sincos(x, &s, &c); return c; }

/* exp(x) */
FHE_FN float fhe_exp(float x)
{
  if (x != x) return x;
  if (x > 88.72283905206835f) return INFINITY;
  if (x < -103.972084045410f) return 0.0f;
  const float n = fhe_rint(x * 1.44269504088896341f);
  float r = fmaf(-n, 0.693359375f, x);
  r = fmaf(-n, -2.12194440e-4f, r);
  float p = fmaf(r, 1.9875691500e-4f, 1.3981999507e-3f);
  p = fmaf(r, p, 8.3334519073e-3f);
  p = fmaf(r, p, 4.1665795894e-2f);
  p = fmaf(r, p, 1.6666665459e-1f);
  p = fmaf(r, p, 5.0000001201e-1f);
  const float e = fmaf(r * r, p, r) + 1.0f;
  /* scale by 2^n in two steps so that subnormal results stay exact-ish */
  const int ni = (int)n;
  const int n1 = ni / 2, n2 = ni - n1;
  const float f1 = fhe_u2f((uint32_t)(n1 + 127) << 23);
  const float f2 = fhe_u2f((uint32_t)(n2 + 127) << 23);
  return (e * f1) * f2;
}

/* ---- double-precision helpers used by pow/log (fp64 +,*,fma are IEEE on both sides) ---- */

/* log2(x) for finite x > 0, ~1e-12 relative */
FHE_FN double fhe_log2_d(double x)
{
  uint64_t u = fhe_d2u(x);
  int e = (int)((u >> 52) & 0x7ff);
  if (e == 0) { /* subnormal double: cannot come from a normal float, but be safe */
    x *= 4503599627370496.0;
    u = fhe_d2u(x);
    e = (int)((u >> 52) & 0x7ff) - 52;
  }
  e -= 1023;
  double m = fhe_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  /* ln(m) = 2 atanh(t), t = (m-1)/(m+1), |t| <= 0.1716 */
  const double t = (m - 1.0) / (m + 1.0);
  const double t2 = t * t;
  double p = 1.0 / 19.0;
  p = fma(p, t2, 1.0 / 17.0);
  p = fma(p, t2, 1.0 / 15.0);
  p = fma(p, t2, 1.0 / 13.0);
  p = fma(p, t2, 1.0 / 11.0);
  p = fma(p, t2, 1.0 / 9.0);
  p = fma(p, t2, 1.0 / 7.0);
  p = fma(p, t2, 1.0 / 5.0);
  p = fma(p, t2, 1.0 / 3.0);
  p = fma(p, t2, 1.0);
  const double ln_m = 2.0 * t * p;
  return fma(ln_m, 1.4426950408889634, (double)e);
}

/* 2^y for |y| < 1000, ~1e-13 relative */
FHE_FN double fhe_exp2_d(double y)
{
  const double magic = 6755399441055744.0; /* 1.5*2^52 */
  const double n = (y + magic) - magic;
  const double r = (y - n) * 0.6931471805599453; /* |r| <= 0.3466 */
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int64_t ni = (int64_t)n;
  return p * fhe_u2d((uint64_t)(ni + 1023) << 52);
}

/* log(x), natural */
FHE_FN float fhe_log(float x)
{
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  return (float)(fhe_log2_d((double)x) * 0.6931471805599453);
}

FHE_FN float fhe_log2(float x)
{
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  return (float)fhe_log2_d((double)x);
}

/* pow(x, y) with the IEEE special cases the integrator can reach */
FHE_FN float fhe_pow(float x, float y)
{
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (x != x || y != y) return NAN;
  if (x == 0.0f) return (y > 0.0f) ? 0.0f : INFINITY;
  if (x < 0.0f) {
    /* negative base: only integer exponents are real */
    const float yi = truncf(y);
    if (yi != y) return NAN;
    const float m = fhe_pow(-x, y);
    return (fmodf(yi, 2.0f) != 0.0f) ? -m : m;
  }
  if (x == INFINITY) return (y > 0.0f) ? INFINITY : 0.0f;
  if (y == INFINITY) return (x > 1.0f) ? INFINITY : 0.0f;
  if (y == -INFINITY) return (x > 1.0f) ? 0.0f : INFINITY;
  const double l = fhe_log2_d((double)x) * (double)y;
  if (l > 128.5) return INFINITY;
  if (l < -151.0) return 0.0f;
  return (float)fhe_exp2_d(l);
}

/* pow(x, 1.5f) as x * sqrt(x): two correctly rounded operations (<= 1.5 ulp of the true power, as good as the general routine above)
 * at a tenth of its cost.  x < 0 gives NaN like pow; +0 gives 0; +inf gives +inf. */
FHE_FN float fhe_pow1p5(float x) { return x * fhe_sqrt(x); }

/* asin kernel for |x| <= 0.5 (cephes asinf) */
FHE_FN float fhe_asin_small(float x)
{
  const float z = x * x;
  float p = fmaf(z, 4.2163199048e-2f, 2.4181311049e-2f);
  p = fmaf(z, p, 4.5470025998e-2f);
  p = fmaf(z, p, 7.4953002686e-2f);
  p = fmaf(z, p, 1.6666752422e-1f);
  return fmaf(p * z, x, x);
}

FHE_FN float fhe_acos(float x)
{
  if (x != x) return x;
  if (x > 1.0f || x < -1.0f) return NAN;
  if (x > 0.5f) return 2.0f * fhe_asin_small(fhe_sqrt(0.5f * (1.0f - x)));
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * fhe_asin_small(fhe_sqrt(0.5f * (1.0f + x)));
  return 1.57079632679489661923f - fhe_asin_small(x);
}

/* atan for x >= 0 (cephes atanf) */
FHE_FN float fhe_atan_pos(float x)
{
  float y0;
  if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
  else y0 = 0.0f;
  const float z = x * x;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(z, p, 1.99777106478e-1f);
  p = fmaf(z, p, -3.33329491539e-1f);
  return y0 + fmaf(p * z, x, x);
}

FHE_FN float fhe_atan2(float y, float x)
{
  if (x != x || y != y) return NAN;
  const float pi = 3.14159265358979323846f;
  if (x == 0.0f) {
    if (y == 0.0f) return (fhe_f2u(x) >> 31) ? copysignf(pi, y) : copysignf(0.0f, y);
    return copysignf(0.5f * pi, y);
  }
  const float a = fhe_atan_pos(fabsf(y / x));
  const float r = (x < 0.0f) ? pi - a : a;
  return copysignf(r, y);
}

#endif /* FH_ELEMENTARY_H */
