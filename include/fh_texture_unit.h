/* fh_texture_unit.h -- normative software texture unit (2-D, normalised coordinates, wrap addressing, bilinear).
 *
 * The reference samples every texture through CUDA texture objects created in cwl/include/cwl/texture.h:35-47
 * (cudaAddressModeWrap on both axes, cudaFilterModeLinear, cudaReadModeNormalizedFloat for uchar4 texels,
 * sRGB -> linear conversion in hardware for COLOR textures, normalizedCoords = 1) and reads them with
 * tex2D<float4>() at 19 sites of fredholm/modules/pt.cu.  gfx950 has no texture path for HIP, and the exact
 * arithmetic of NVIDIA's texture unit is not in the tree, so this header fixes ONE definition following the
 * CUDA Programming Guide's description of linear filtering ("Texture Fetching"): texel centres at +0.5,
 * the two weights held in 1.8 fixed point (8 fractional bits), sRGB decoded per texel BEFORE filtering.
 * Both the HIP kernels and the CPU checker include it, so their results agree bit for bit; agreement
 * with NVIDIA hardware beyond the documented behaviour is unpinned.
 */
#ifndef FH_TEXTURE_UNIT_H
#define FH_TEXTURE_UNIT_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FHT_FN __host__ __device__ inline
#else
#define FHT_FN static inline
#endif

typedef struct fht_texture {
  const uint8_t* rgba8;  /* width*height*4 bytes, row-major, row 0 first; NULL for float textures */
  const float* rgba32f;  /* width*height*4 floats (IBL); NULL for 8-bit textures */
  uint32_t width, height;
  uint32_t srgb;         /* 1: r,g,b bytes are sRGB-encoded (COLOR textures); alpha is always linear */
} fht_texture;

/* 256-entry sRGB EOTF table, filled once on the host with fht_srgb_to_linear */
FHT_FN float fht_srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : (float)pow(((double)c + 0.055) / 1.055, 2.4); }

FHT_FN int fht_wrap(int i, int n)
{
  i %= n;
  return i < 0 ? i + n : i;
}

FHT_FN void fht_texel(const fht_texture* t, const float* srgb_lut, int x, int y, float out[4])
{
  const size_t k = ((size_t)fht_wrap(y, (int)t->height) * t->width + (size_t)fht_wrap(x, (int)t->width)) * 4u;
  if (t->rgba32f) {
    out[0] = t->rgba32f[k]; out[1] = t->rgba32f[k + 1]; out[2] = t->rgba32f[k + 2]; out[3] = t->rgba32f[k + 3];
    return;
  }
  const uint8_t* p = t->rgba8 + k;
  if (t->srgb) { out[0] = srgb_lut[p[0]]; out[1] = srgb_lut[p[1]]; out[2] = srgb_lut[p[2]]; }
  else { out[0] = p[0] * (1.0f / 255.0f); out[1] = p[1] * (1.0f / 255.0f); out[2] = p[2] * (1.0f / 255.0f); }
  out[3] = p[3] * (1.0f / 255.0f);
}

/* tex2D<float4>(tex, u, v) */
FHT_FN void fht_tex2d(const fht_texture* t, const float* srgb_lut, float u, float v, float out[4])
{
  if (!(u == u) || !(v == v) || t->width == 0 || t->height == 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
  /* wrap: keep the fractional part of the normalised coordinate */
  u = u - floorf(u);
  v = v - floorf(v);
  const float xb = u * (float)t->width - 0.5f, yb = v * (float)t->height - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb);
  /* 1.8 fixed-point weights */
  const float a = floorf((xb - xf) * 256.0f + 0.5f) * (1.0f / 256.0f);
  const float b = floorf((yb - yf) * 256.0f + 0.5f) * (1.0f / 256.0f);
  const int i = (int)xf, j = (int)yf;
  float t00[4], t10[4], t01[4], t11[4];
  fht_texel(t, srgb_lut, i, j, t00);
  fht_texel(t, srgb_lut, i + 1, j, t10);
  fht_texel(t, srgb_lut, i, j + 1, t01);
  fht_texel(t, srgb_lut, i + 1, j + 1, t11);
  for (int c = 0; c < 4; ++c)
    out[c] = (1.0f - a) * (1.0f - b) * t00[c] + a * (1.0f - b) * t10[c] + (1.0f - a) * b * t01[c] + a * b * t11[c];
}

#endif /* FH_TEXTURE_UNIT_H */
