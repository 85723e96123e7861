/* fh_texture_unit.h -- normative software texture unit (2-D, normalised coordinates, wrap addressing, bilinear).
 *
 * The reference samples every texture through CUDA texture objects created in cwl/include/cwl/texture.h:35-47
 * (cudaAddressModeWrap on both axes, cudaFilterModeLinear, cudaReadModeNormalizedFloat for uchar4 texels,
 * sRGB -> linear conversion in hardware for COLOR textures, normalizedCoords = 1) and reads them with
 * tex2D<float4>() at 19 sites of fredholm/modules/pt.cu.  gfx950 has no texture path for HIP, and the exact
 * arithmetic of NVIDIA's texture unit is not in the tree, so this header fixes ONE definition following the
 * CUDA Programming Guide's description of linear filtering ("Texture Fetching"): texel centres at +0.5,
 * the two weights held in 1.8 fixed point (8 fractional bits), sRGB decoded per texel BEFORE filtering;
 * a NaN or infinite coordinate fetches 0 in every channel.
 * The CPU checker has its own implementation of the same definition (oracle/otexture.h) and the two are compared texel for texel
 * (fh_kat_tex2d); agreement with NVIDIA hardware beyond the documented behaviour is unpinned.
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

/* Addressing of one fetch: the four texel coordinates (wrapped) and the two 1.8 fixed-point weights.  After the wrap the normalised coordinate is in
 * [0, 1], so the unwrapped texel index floor(u * n - 0.5) lies in [-1, n - 1] and wrapping it (and its successor) needs one compare each instead of a
 * remainder.  Returns 0 for a coordinate that is NaN or infinite (and for an empty texture): the fetch then yields 0 in every channel. */
typedef struct fht_address {
  int i0, i1, j0, j1; /* columns i, i+1 and rows j, j+1, wrapped into the texture */
  float a, b;         /* weight of column i+1 / of row j+1 */
} fht_address;

FHT_FN int fht_wrap_low(int i, int n) { return i < 0 ? i + n : i; }   /* i in [-1, n-1] */
FHT_FN int fht_wrap_high(int i, int n) { return i >= n ? i - n : i; } /* i in [0, n] */

FHT_FN int fht_address_of(uint32_t width, uint32_t height, float u, float v, fht_address* ad)
{
  if (!(fabsf(u) <= 3.402823466e38f) || !(fabsf(v) <= 3.402823466e38f) || width == 0 || height == 0) return 0;
  /* wrap: keep the fractional part of the normalised coordinate */
  u = u - floorf(u);
  v = v - floorf(v);
  const float xb = u * (float)width - 0.5f, yb = v * (float)height - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb);
  /* 1.8 fixed-point weights */
  ad->a = floorf((xb - xf) * 256.0f + 0.5f) * (1.0f / 256.0f);
  ad->b = floorf((yb - yf) * 256.0f + 0.5f) * (1.0f / 256.0f);
  const int i = (int)xf, j = (int)yf;
  ad->i0 = fht_wrap_low(i, (int)width);
  ad->i1 = fht_wrap_high(i + 1, (int)width);
  ad->j0 = fht_wrap_low(j, (int)height);
  ad->j1 = fht_wrap_high(j + 1, (int)height);
  return 1;
}

FHT_FN float fht_blend(const fht_address* ad, float t00, float t10, float t01, float t11)
{
  const float a = ad->a, b = ad->b;
  return (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
}

/* texel (x, y), both already inside the texture */
FHT_FN void fht_texel(const fht_texture* t, const float* srgb_lut, int x, int y, float out[4])
{
  const size_t k = ((size_t)y * t->width + (size_t)x) * 4u;
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
  fht_address ad;
  if (!fht_address_of(t->width, t->height, u, v, &ad)) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
  float t00[4], t10[4], t01[4], t11[4];
  fht_texel(t, srgb_lut, ad.i0, ad.j0, t00);
  fht_texel(t, srgb_lut, ad.i1, ad.j0, t10);
  fht_texel(t, srgb_lut, ad.i0, ad.j1, t01);
  fht_texel(t, srgb_lut, ad.i1, ad.j1, t11);
  for (int c = 0; c < 4; ++c) out[c] = fht_blend(&ad, t00[c], t10[c], t01[c], t11[c]);
}

/* one channel of tex2D<float4>() on an 8-bit texture: the same value as fht_tex2d(...)[channel] from a quarter of the texel reads.
 * `decode` = the 256-entry sRGB table for an r, g or b channel of a COLOR texture, NULL for a linear channel (alpha always is). */
FHT_FN float fht_tex2d_channel8(const uint8_t* rgba8, uint32_t width, uint32_t height, const float* decode, uint32_t channel, float u, float v)
{
  fht_address ad;
  if (!fht_address_of(width, height, u, v, &ad)) return 0.0f;
  const uint8_t b00 = rgba8[((size_t)ad.j0 * width + (size_t)ad.i0) * 4u + channel], b10 = rgba8[((size_t)ad.j0 * width + (size_t)ad.i1) * 4u + channel];
  const uint8_t b01 = rgba8[((size_t)ad.j1 * width + (size_t)ad.i0) * 4u + channel], b11 = rgba8[((size_t)ad.j1 * width + (size_t)ad.i1) * 4u + channel];
  if (decode) return fht_blend(&ad, decode[b00], decode[b10], decode[b01], decode[b11]);
  return fht_blend(&ad, b00 * (1.0f / 255.0f), b10 * (1.0f / 255.0f), b01 * (1.0f / 255.0f), b11 * (1.0f / 255.0f));
}

#endif /* FH_TEXTURE_UNIT_H */
