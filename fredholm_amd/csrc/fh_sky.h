// fh_sky.h -- Hosek-Wilkie RGB sky: host-side coefficient cook + device radiance evaluation.
// Behavioural source: fredholm/include/fredholm/arhosek.h:145-322 (cook; quintic Bezier in
// cbrt(elevation / (pi/2)), bilinear in turbidity and albedo), fredholm/modules/arhosek.cu:103-127
// (radiance), fredholm/modules/pt.cu:352-363 (theta/gamma from the direction).
// Only the 3 RGB channels of the reference's 11-channel state are ever read (arhosek.h:313-320).
#pragma once
#include "../../include/fh_elementary.h"
#include "fh_vec.h"

namespace fh {

struct HosekSky { float cfg[3][9]; float rad[3]; };

inline float hosek_bezier(const float* m, int stride, float e)
{
  const float ie = 1.0f - e;
  return fhe_pow(ie, 5.0f) * m[0] + 5.0f * fhe_pow(ie, 4.0f) * e * m[stride] + 10.0f * fhe_pow(ie, 3.0f) * fhe_pow(e, 2.0f) * m[2 * stride] +
         10.0f * fhe_pow(ie, 2.0f) * fhe_pow(e, 3.0f) * m[3 * stride] + 5.0f * ie * fhe_pow(e, 4.0f) * m[4 * stride] + fhe_pow(e, 5.0f) * m[5 * stride];
}

// table = 3 x 1080 config floats followed by 3 x 120 radiance floats
inline HosekSky hosek_cook(const float* table, float turbidity, float albedo, float elevation)
{
  HosekSky st{};
  const int it = (int)turbidity;
  const float tr = turbidity - (float)it;
  const float e = fhe_pow(elevation / (kPi / 2.0f), (1.0f / 3.0f));
  for (int ch = 0; ch < 3; ++ch) {
    const float* ds = table + 1080 * ch;
    const float* dr = table + 3240 + 120 * ch;
    for (int i = 0; i < 9; ++i) {
      float c = (1.0f - albedo) * (1.0f - tr) * hosek_bezier(ds + 54 * (it - 1) + i, 9, e);
      c += albedo * (1.0f - tr) * hosek_bezier(ds + 540 + 54 * (it - 1) + i, 9, e);
      if (it != 10) {
        c += (1.0f - albedo) * tr * hosek_bezier(ds + 54 * it + i, 9, e);
        c += albedo * tr * hosek_bezier(ds + 540 + 54 * it + i, 9, e);
      }
      st.cfg[ch][i] = c;
    }
    float r = (1.0f - albedo) * (1.0f - tr) * hosek_bezier(dr + 6 * (it - 1), 1, e);
    r += albedo * (1.0f - tr) * hosek_bezier(dr + 60 + 6 * (it - 1), 1, e);
    if (it != 10) {
      r += (1.0f - albedo) * tr * hosek_bezier(dr + 6 * it, 1, e);
      r += albedo * tr * hosek_bezier(dr + 60 + 6 * it, 1, e);
    }
    st.rad[ch] = r;
  }
  return st;
}

// the three channels share every transcendental that does not depend on the coefficients
FH_HD f3 hosek_radiance(const HosekSky& st, f3 sun_dir, float intensity, f3 v)
{
  const float theta = fhe_acos(clampf(v.y, -1.0f, 1.0f));
  const float gamma = fhe_acos(dot(sun_dir, v));
  const float cg = fhe_cos(gamma), ct = fhe_cos(theta);
  const float rayM = cg * cg;
  const float zenith = sqrt_cr(ct);
  float out[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* c = st.cfg[ch];
    const float expM = fhe_exp(c[4] * gamma);
    const float mieM = (1.0f + cg * cg) / fhe_pow1p5(1.0f + c[8] * c[8] - 2.0f * c[8] * cg);  // arhosek.cu:109-110: pow(x, 1.5)
    out[ch] = (1.0f + c[0] * fhe_exp(c[1] / (ct + 0.01f))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith) * st.rad[ch];
  }
  return intensity * mk3(out[0], out[1], out[2]);
}

}  // namespace fh
