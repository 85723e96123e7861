// otexture.h -- TEST INFRASTRUCTURE: the checker's OWN texture unit, written from the stated definition and not from the product's
// include/fh_texture_unit.h (which it does not include), so that product-vs-checker agreement on textured scenes is a check and not a tautology.
//
// Definition (CUDA Programming Guide, "Texture Fetching", linear filtering, as the reference configures its texture objects in
// cwl/include/cwl/texture.h:35-47: wrap addressing on both axes, normalised coordinates, cudaFilterModeLinear, uchar4 -> normalised float,
// sRGB -> linear per texel for COLOR textures):
//   x_B = frac(u) * W - 0.5,  i = floor(x_B),  alpha = frac(x_B) stored in 9-bit fixed point with 8 fractional bits (round to nearest)
//   tex = (1-alpha)(1-beta) T[i,j] + alpha (1-beta) T[i+1,j] + (1-alpha) beta T[i,j+1] + alpha beta T[i+1,j+1],  indices wrapped
// The weights are formed as integers here (0..256) and the texel decode goes through tables; the final blend is the same four-term float sum.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc {

struct OTexture {
  const uint8_t* rgba8 = nullptr;  // W*H*4, row 0 first
  const float* rgba32f = nullptr;  // float4 texels (IBL), or nullptr
  uint32_t width = 0, height = 0;
  bool srgb = false;
};

struct TexelTables {  // byte -> float, built once
  float linear[256], srgb[256];
  TexelTables()
  {
    for (int b = 0; b < 256; ++b) {
      const float c = (float)b * (1.0f / 255.0f);
      linear[b] = c;
      srgb[b] = c <= 0.04045f ? c / 12.92f : (float)std::pow(((double)c + 0.055) / 1.055, 2.4);  // IEC 61966-2-1 EOTF
    }
  }
};
inline const TexelTables& texel_tables() { static const TexelTables t; return t; }

inline uint32_t wrap_index(long long i, uint32_t n)
{
  long long m = i % (long long)n;
  if (m < 0) m += n;
  return (uint32_t)m;
}

inline void fetch_texel(const OTexture& t, long long x, long long y, float out[4])
{
  const size_t k = ((size_t)wrap_index(y, t.height) * t.width + wrap_index(x, t.width)) * 4u;
  if (t.rgba32f) { for (int c = 0; c < 4; ++c) out[c] = t.rgba32f[k + c]; return; }
  const TexelTables& tb = texel_tables();
  const float* rgb = t.srgb ? tb.srgb : tb.linear;
  out[0] = rgb[t.rgba8[k]]; out[1] = rgb[t.rgba8[k + 1]]; out[2] = rgb[t.rgba8[k + 2]];
  out[3] = tb.linear[t.rgba8[k + 3]];  // alpha is never sRGB-encoded
}

// one axis: integer texel index and the 1.8 fixed-point weight (0..256) of the NEXT texel
inline void axis(float coord, uint32_t n, long long& index, int& weight256)
{
  const float frac = coord - std::floor(coord);          // wrap addressing keeps the fractional part
  const float pos = frac * (float)n - 0.5f;
  const float base = std::floor(pos);
  index = (long long)base;
  weight256 = (int)std::floor((pos - base) * 256.0f + 0.5f);
}

// tex2D<float4>(texture, u, v)
inline void tex2d(const OTexture& t, float u, float v, float out[4])
{
  if (!std::isfinite(u) || !std::isfinite(v) || t.width == 0 || t.height == 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }  // a NaN or infinite coordinate fetches 0
  long long i, j;
  int wa, wb;
  axis(u, t.width, i, wa);
  axis(v, t.height, j, wb);
  const float a = (float)wa * (1.0f / 256.0f), b = (float)wb * (1.0f / 256.0f);
  float t00[4], t10[4], t01[4], t11[4];
  fetch_texel(t, i, j, t00);
  fetch_texel(t, i + 1, j, t10);
  fetch_texel(t, i, j + 1, t01);
  fetch_texel(t, i + 1, j + 1, t11);
  for (int c = 0; c < 4; ++c) out[c] = (1.0f - a) * (1.0f - b) * t00[c] + a * (1.0f - b) * t10[c] + (1.0f - a) * b * t01[c] + a * b * t11[c];
}

}  // namespace orc
