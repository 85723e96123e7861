// fh_tonemap.h -- per-pixel helpers of the post-process chain (host + device), shared by post.hip and the KAT entry points.
// Restates fredholm/kernels/include/kernels/post-process.h: smoothstep :53-64, uchimura :78-111, linear_to_srgb :19-29,
// compute_EV100 :114-119, convert_EV100_to_exposure :121-125, rgb_to_luminance :13-16.  Mixed float/double arithmetic follows the
// reference's literals (12.92, 1.055, 100.0, 1.2 are doubles there).  Pinned against the reference's own header, built for the
// host into oracle/_ref/libref_lut_math_post.so (tests/golden/ref_post_*.npz).
#pragma once
#include "../../include/fh_elementary.h"

#if defined(__HIPCC__)
#define FH_TM __host__ __device__ __forceinline__
#else
#define FH_TM inline
#endif

namespace fh {

FH_TM float luminance_rgb(float r, float g, float b) { return r * 0.2126729f + g * 0.7151522f + b * 0.0721750f; }
FH_TM float smoothstep_f(float e0, float e1, float x)
{
  if (x < e0) return 0.0f;
  if (x > e1) return 1.0f;
  x = (x - e0) / (e1 - e0);
  return x * x * (3.0f - 2.0f * x);
}
FH_TM float uchimura1(float x)
{
  const float P = 1.0f, a = 1.0f, m = 0.22f, l = 0.4f, c = 1.33f, b = 0.0f;
  const float l0 = ((P - m) * l) / a;
  const float S0 = m + l0;
  const float S1 = m + a * l0;
  const float C2 = (a * P) / (P - S1);
  const float CP = -C2 / P;
  const float w0 = 1.0f - smoothstep_f(0.0f, m, x);
  const float w2 = (x < m + l0) ? 0.0f : 1.0f;
  const float w1 = 1.0f - w0 - w2;
  const float T = m * fhe_pow(x / m, c) + b;
  const float S = P - (P - S1) * fhe_exp(CP * (x - S0));
  const float Lc = m + a * (x - m);
  return T * w0 + Lc * w1 + S * w2;
}
FH_TM float srgb1(float x) { return x < 0.0031308 ? (float)(12.92 * x) : (float)(1.055 * fhe_pow(x, 1.0f / 2.4f) - 0.055); }
FH_TM float clamp01f(float v) { return fmaxf(0.0f, fminf(v, 1.0f)); }
FH_TM float ev100_of(float aperture, float shutter, float iso) { return fhe_log2((float)(aperture * aperture / shutter * 100.0 / iso)); }
FH_TM float exposure_from_ev100(float ev100)
{
  const float max_luminance = (float)(1.2 * fhe_pow(2.0f, ev100));
  return 1.0f / max_luminance;
}

}  // namespace fh
