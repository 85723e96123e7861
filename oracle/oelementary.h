// oelementary.h -- the CPU checker's OWN elementary functions (test infrastructure; nothing in the product includes this file).
//
// The reference calls CUDA's libm on the device (sinf / cosf in sampling.cu:54-64, expf / powf / acosf in arhosek.cu:103-118, powf in bxdf.cu:784-791,
// log2f / powf in kernels/post-process.h:96-124).  CUDA's roundings are not available here, glibc's differ from ROCm's, and a path tracer turns a last-bit
// difference into a different hit triangle, so product and checker agree on ONE numerical definition of these functions ("numerical
// specification"): fp32 results built from IEEE-754 +, -, x, /, sqrt, fma and integer bit operations only, which round identically on x86-64 and gfx950.
// The product's implementation is include/fh_elementary.h; this file is a second implementation written from the specification -- polynomial coefficients in
// tables, one Horner routine, its own handling of the special cases -- and shares no code with it.  tests/test_oracle_anchors.py compiles the product's header
// for the host and requires both to return the same bits over millions of arguments, and checks both against float64 libm.
//
// Specification (all arithmetic fp32 unless marked fp64; fma = fused multiply-add; rn(x) = (x + 1.5 * 2^23) - 1.5 * 2^23, round to nearest even for |x| < 2^22):
//   sincos(x): k = rn(x * 2/pi); r = fma(-k, P3, fma(-k, P2, fma(-k, P1, x))) with pi/2 = P1 + P2 + P3 (Cody-Waite);  z = r * r;
//              S = fma(z * r, horner(z; S2, S1, S0), r);  C = fma(z * z, horner(z; C2, C1, C0), fma(z, -1/2, 1));  quadrant q = int(k) & 3 swaps / negates.
//   exp(x):    NaN -> NaN, x > 88.72283905206835 -> inf, x < -103.972084045410 -> 0;  n = rn(x * log2(e));  r = fma(-n, L2, fma(-n, L1, x)) with ln 2 = L1 + L2;
//              e = fma(r * r, horner(r; E5 .. E0), r) + 1;  result = (e * 2^(n div 2)) * 2^(n - n div 2), div truncating.
//   log2_d(x) (fp64): x = m * 2^e, m in [1, 2); m > sqrt 2 -> m / 2, e + 1;  t = (m - 1) / (m + 1);  ln m = 2 t horner(t^2; 1/19, 1/17, .. 1/3, 1);  fma(ln m, log2(e), e).
//   exp2_d(y) (fp64): n = (y + 1.5 * 2^52) - 1.5 * 2^52;  r = (y - n) ln 2;  horner(r; 1/12!, 1/11!, .. 1/2!, 1, 1) * 2^n.
//   log(x) = float(log2_d(x) * ln 2), log2(x) = float(log2_d(x)) with NaN -> NaN, x < 0 -> NaN, 0 -> -inf, inf -> inf.
//   pow(x, y): IEEE special cases, then float(exp2_d(log2_d(x) * y)) with overflow above 128.5 and underflow below -151;  pow1p5(x) = x * sqrt(x).
//   acos(x):   |x| <= 1/2: pi/2 - A(x);  x > 1/2: 2 A(sqrt((1 - x) / 2));  x < -1/2: pi - 2 A(sqrt((1 + x) / 2));  A(x) = fma(horner(z; A4 .. A0) * z, x, x), z = x * x.
//   atan2(y, x): reduction of |y / x| at tan(3 pi / 8) and tan(pi / 8), T(x) = fma(horner(z; T3 .. T0) * z, x, x), quadrant from the signs.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace oe {

template <class To, class From>
inline To bits(From v)
{
  static_assert(sizeof(To) == sizeof(From), "same width");
  To r;
  std::memcpy(&r, &v, sizeof r);
  return r;
}

// c[0] * x^(N-1) + ... + c[N-1], every step one fused multiply-add
template <class T, int N>
inline T horner(T x, const T (&c)[N])
{
  T p = c[0];
  for (int i = 1; i < N; ++i) p = std::fma(p, x, c[i]);
  return p;
}

inline float round_even(float x)
{
  const float shift = 12582912.0f;
  volatile float t = x + shift;  // (volatile: the two roundings must both happen, whatever the optimiser thinks of x + c - c)
  return t - shift;
}

struct SinCos { float s, c; };
inline SinCos sincos(float x)
{
  static const float kSin[3] = {-1.9515295891e-4f, 8.3321608736e-3f, -1.6666654611e-1f};
  static const float kCos[3] = {2.443315711809948e-5f, -1.388731625493765e-3f, 4.166664568298827e-2f};
  static const float kPio2[3] = {1.5703125f, 4.83751296997070312e-4f, 7.54978995489188216e-8f};
  const float k = round_even(x * 0.636619772367581343f);
  float r = x;
  for (float part : kPio2) r = std::fma(-k, part, r);
  const float z = r * r;
  const float sin_r = std::fma(z * r, horner(z, kSin), r);
  const float cos_r = std::fma(z * z, horner(z, kCos), std::fma(z, -0.5f, 1.0f));
  SinCos out;
  switch (static_cast<int>(k) & 3) {
    case 0: out.s = sin_r; out.c = cos_r; break;
    case 1: out.s = cos_r; out.c = -sin_r; break;
    case 2: out.s = -sin_r; out.c = -cos_r; break;
    default: out.s = -cos_r; out.c = sin_r; break;
  }
  return out;
}
inline float sin(float x) { return sincos(x).s; }
inline float cos(float x) { return sincos(x).c; }

inline float pow2i(int n) { return bits<float>(static_cast<uint32_t>(n + 127) << 23); }

inline float exp(float x)
{
  static const float kExp[6] = {1.9875691500e-4f, 1.3981999507e-3f, 8.3334519073e-3f, 4.1665795894e-2f, 1.6666665459e-1f, 5.0000001201e-1f};
  if (std::isnan(x)) return x;
  if (x > 88.72283905206835f) return std::numeric_limits<float>::infinity();
  if (x < -103.972084045410f) return 0.0f;
  const float n = round_even(x * 1.44269504088896341f);
  const float r = std::fma(-n, -2.12194440e-4f, std::fma(-n, 0.693359375f, x));
  const float e = std::fma(r * r, horner(r, kExp), r) + 1.0f;
  const int ni = static_cast<int>(n), half = ni / 2;
  return (e * pow2i(half)) * pow2i(ni - half);
}

inline double log2_d(double x)
{
  static const double kAtanh[10] = {1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0};
  uint64_t u = bits<uint64_t>(x);
  int e = static_cast<int>((u >> 52) & 0x7ffu);
  if (e == 0) {  // subnormal double (cannot come from a normal float)
    x *= 4503599627370496.0;
    u = bits<uint64_t>(x);
    e = static_cast<int>((u >> 52) & 0x7ffu) - 52;
  }
  e -= 1023;
  double m = bits<double>((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > 1.4142135623730951) { m *= 0.5; ++e; }
  const double t = (m - 1.0) / (m + 1.0);
  const double ln_m = 2.0 * t * horner(t * t, kAtanh);
  return std::fma(ln_m, 1.4426950408889634, static_cast<double>(e));
}

inline double exp2_d(double y)
{
  static const double kInvFact[13] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
                                      1.0 / 120.0,       1.0 / 24.0,       1.0 / 6.0,       0.5,            1.0,           1.0};
  const double shift = 6755399441055744.0;
  volatile double t = y + shift;
  const double n = t - shift;
  const double r = (y - n) * 0.6931471805599453;
  return horner(r, kInvFact) * bits<double>(static_cast<uint64_t>(static_cast<int64_t>(n) + 1023) << 52);
}

// what log / log2 return without computing anything: NaN for NaN and negative arguments, -inf at 0, +inf at +inf
inline bool log_special(float x, float& out)
{
  if (std::isnan(x)) { out = x; return true; }
  if (x < 0.0f) { out = std::numeric_limits<float>::quiet_NaN(); return true; }
  if (x == 0.0f) { out = -std::numeric_limits<float>::infinity(); return true; }
  if (std::isinf(x)) { out = x; return true; }
  return false;
}
inline float log(float x) { float s; return log_special(x, s) ? s : static_cast<float>(log2_d(static_cast<double>(x)) * 0.6931471805599453); }
inline float log2(float x) { float s; return log_special(x, s) ? s : static_cast<float>(log2_d(static_cast<double>(x))); }

inline float pow(float x, float y)
{
  const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
  if (y == 0.0f || x == 1.0f) return 1.0f;
  if (std::isnan(x) || std::isnan(y)) return nan;
  if (x == 0.0f) return y > 0.0f ? 0.0f : inf;
  if (x < 0.0f) {  // real only for integer exponents; odd ones keep the sign
    if (std::trunc(y) != y) return nan;
    const float mag = pow(-x, y);
    return std::fmod(y, 2.0f) != 0.0f ? -mag : mag;
  }
  if (x == inf) return y > 0.0f ? inf : 0.0f;
  if (std::isinf(y)) return ((x > 1.0f) == (y > 0.0f)) ? inf : 0.0f;
  const double l = log2_d(static_cast<double>(x)) * static_cast<double>(y);
  if (l > 128.5) return inf;
  if (l < -151.0) return 0.0f;
  return static_cast<float>(exp2_d(l));
}
inline float pow1p5(float x) { return x * std::sqrt(x); }

inline float asin_core(float x)
{
  static const float kAsin[5] = {4.2163199048e-2f, 2.4181311049e-2f, 4.5470025998e-2f, 7.4953002686e-2f, 1.6666752422e-1f};
  const float z = x * x;
  return std::fma(horner(z, kAsin) * z, x, x);
}
inline float acos(float x)
{
  if (std::isnan(x)) return x;
  if (std::fabs(x) > 1.0f) return std::numeric_limits<float>::quiet_NaN();
  if (x > 0.5f) return 2.0f * asin_core(std::sqrt(0.5f * (1.0f - x)));
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin_core(std::sqrt(0.5f * (1.0f + x)));
  return 1.57079632679489661923f - asin_core(x);
}

inline float atan_nonneg(float x)
{
  static const float kAtan[4] = {8.05374449538e-2f, -1.38776856032e-1f, 1.99777106478e-1f, -3.33329491539e-1f};
  float base = 0.0f;
  if (x > 2.414213562373095f) { base = 1.57079632679489661923f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { base = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
  const float z = x * x;
  return base + std::fma(horner(z, kAtan) * z, x, x);
}
inline float atan2(float y, float x)
{
  const float pi = 3.14159265358979323846f;
  if (std::isnan(x) || std::isnan(y)) return std::numeric_limits<float>::quiet_NaN();
  if (x == 0.0f) {
    if (y == 0.0f) return std::copysign(std::signbit(x) ? pi : 0.0f, y);
    return std::copysign(0.5f * pi, y);
  }
  const float a = atan_nonneg(std::fabs(y / x));
  return std::copysign(x < 0.0f ? pi - a : a, y);
}

}  // namespace oe
