// fh_sampler.h -- device samplers: position-keyed CMJ (2-D) and Owen-scrambled Sobol' (1-D).
//
// The reference carries a 72-byte sampler state per path (shared.h:66-96) and advances two
// counters as it draws (cmj.cu:78, sobol.cu:10740).  In a wavefront integrator both counters are
// pure functions of (bounce, which light types are enabled), so no state is stored per path:
// every draw is addressed by its absolute slot and the keys are recomputed from
// (pixel, sample index, slot, seed).  Integer results are bit-exact with
//   xxhash32            shared.h:282-319
//   cmj_permute / cmj   cmj.cu:12-69        (M = N = 4: the rejection loop runs exactly once)
//   Sobol' + Owen       sobol.cu:10661-10742 (32-bit truncation of the index included)
#pragma once
#include "../../include/fh_elementary.h"
#include "fh_vec.h"

namespace fh {

FH_HD uint32_t rotl17(uint32_t v) { return (v << 17) | (v >> 15); }

FH_HD uint32_t xxhash32(uint32_t p)
{
  uint32_t h = p + 374761393U;
  h = 668265263U * rotl17(h);
  h = 2246822519U * (h ^ (h >> 15));
  h = 3266489917U * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
FH_HD uint32_t xxhash32(uint32_t x, uint32_t y, uint32_t z)
{
  uint32_t h = z + 374761393U + x * 3266489917U;
  h = 668265263U * rotl17(h);
  h += y * 3266489917U;
  h = 668265263U * rotl17(h);
  h = 2246822519U * (h ^ (h >> 15));
  h = 3266489917U * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
FH_HD uint32_t xxhash32(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
  uint32_t h = w + 374761393U + x * 3266489917U;
  h = 668265263U * rotl17(h);
  h += y * 3266489917U;
  h = 668265263U * rotl17(h);
  h += z * 3266489917U;
  h = 668265263U * rotl17(h);
  h = 2246822519U * (h ^ (h >> 15));
  h = 3266489917U * (h ^ (h >> 13));
  return h ^ (h >> 16);
}

// Kensler's permutation for a power-of-two length l (mask w = l-1): one round, no rejection.
template <uint32_t L>
FH_HD uint32_t cmj_permute_pow2(uint32_t i, uint32_t p)
{
  constexpr uint32_t w = L - 1;
  i ^= p;             i *= 0xe170893d;
  i ^= p >> 16;
  i ^= (i & w) >> 4;
  i ^= p >> 8;        i *= 0x0929eb3f;
  i ^= p >> 23;
  i ^= (i & w) >> 1;  i *= 1 | p >> 27;
                      i *= 0x6935fa69;
  i ^= (i & w) >> 11; i *= 0x74dcb303;
  i ^= (i & w) >> 2;  i *= 0x9e501cc3;
  i ^= (i & w) >> 2;  i *= 0xc860a3df;
  i &= w;
  i ^= i >> 5;
  return (i + p) % L;
}
// general-length form, used only by the known-answer entry point
FH_HD uint32_t cmj_permute(uint32_t i, uint32_t l, uint32_t p)
{
  uint32_t w = l - 1;
  w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
  do {
    i ^= p;             i *= 0xe170893d;
    i ^= p >> 16;
    i ^= (i & w) >> 4;
    i ^= p >> 8;        i *= 0x0929eb3f;
    i ^= p >> 23;
    i ^= (i & w) >> 1;  i *= 1 | p >> 27;
                        i *= 0x6935fa69;
    i ^= (i & w) >> 11; i *= 0x74dcb303;
    i ^= (i & w) >> 2;  i *= 0x9e501cc3;
    i ^= (i & w) >> 2;  i *= 0xc860a3df;
    i &= w;
    i ^= i >> 5;
  } while (i >= l);
  return (i + p) % l;
}
FH_HD float cmj_randfloat(uint32_t i, uint32_t p)
{
  i ^= p;
  i ^= i >> 17;
  i ^= i >> 10; i *= 0xb36534e5;
  i ^= i >> 12;
  i ^= i >> 21; i *= 0x93fc4795;
  i ^= 0xdf6e307f;
  i ^= i >> 17; i *= 1 | p >> 18;
  return i * (1.0f / 4294967808.0f);
}

// 2-D draw number `slot` of sample `n_spp` at pixel `image_idx` (cmj.cu:60-80)
FH_HD f2 cmj_draw(uint32_t n_spp, uint32_t image_idx, uint32_t slot, uint32_t seed_hash)
{
  uint32_t index = n_spp % 16u;
  const uint32_t scramble = xxhash32(n_spp / 16u, image_idx, slot, seed_hash);
  index = cmj_permute_pow2<16>(index, scramble * 0x51633e2d);
  const uint32_t sx = cmj_permute_pow2<4>(index % 4u, scramble * 0xa511e9b3);
  const uint32_t sy = cmj_permute_pow2<4>(index / 4u, scramble * 0x63d83595);
  const float jx = cmj_randfloat(index, scramble * 0xa399d265);
  const float jy = cmj_randfloat(index, scramble * 0x711ad6a5);
  return mk2((index % 4u + (sy + jx) / 4) / 4, (index / 4u + (sx + jy) / 4) / 4);
}

// The same draw for a lane that walks the samples of ONE pixel in order (k_sky_pixels): the scramble, its five products and the two permutations of four depend on
// (n_spp / 16, pixel, slot, seed) only, i.e. they are the same for sixteen consecutive samples -- and a permutation of four is a byte (two bits per entry).  What is left
// per sample is the permutation of sixteen and the two jitters: 35 % of cmj_draw's integer work.  Integer for integer and float for float the operations of cmj_draw.
struct CmjBlock { uint32_t p_index, p_jx, p_jy, sx4, sy4; };
FH_HD CmjBlock cmj_block(uint32_t n_spp_div16, uint32_t image_idx, uint32_t slot, uint32_t seed_hash)
{
  const uint32_t scramble = xxhash32(n_spp_div16, image_idx, slot, seed_hash);
  CmjBlock b;
  b.p_index = scramble * 0x51633e2d; b.p_jx = scramble * 0xa399d265; b.p_jy = scramble * 0x711ad6a5;
  b.sx4 = b.sy4 = 0u;
#pragma unroll
  for (uint32_t j = 0; j < 4u; ++j) {
    b.sx4 |= cmj_permute_pow2<4>(j, scramble * 0xa511e9b3) << (2u * j);
    b.sy4 |= cmj_permute_pow2<4>(j, scramble * 0x63d83595) << (2u * j);
  }
  return b;
}
FH_HD f2 cmj_draw_in_block(const CmjBlock& b, uint32_t n_spp)
{
  const uint32_t index = cmj_permute_pow2<16>(n_spp % 16u, b.p_index);
  const uint32_t sx = (b.sx4 >> (2u * (index % 4u))) & 3u;
  const uint32_t sy = (b.sy4 >> (2u * (index / 4u))) & 3u;
  const float jx = cmj_randfloat(index, b.p_jx);
  const float jy = cmj_randfloat(index, b.p_jy);
  return mk2((index % 4u + (sy + jx) / 4) / 4, (index / 4u + (sx + jy) / 4) / 4);
}

FH_HD uint32_t reverse_bits32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x);
#else
  x = ((x & 0xaaaaaaaa) >> 1) | ((x & 0x55555555) << 1);
  x = ((x & 0xcccccccc) >> 2) | ((x & 0x33333333) << 2);
  x = ((x & 0xf0f0f0f0) >> 4) | ((x & 0x0f0f0f0f) << 4);
  x = ((x & 0xff00ff00) >> 8) | ((x & 0x00ff00ff) << 8);
  return (x >> 16) | (x << 16);
#endif
}
FH_HD uint32_t laine_karras(uint32_t x, uint32_t seed)
{
  x += seed;
  x ^= x * 0x6c50b47cu;
  x ^= x * 0xb82f1e52u;
  x ^= x * 0xc7afe638u;
  x ^= x * 0x8d22f6e6u;
  return x;
}
FH_HD uint32_t owen_scramble(uint32_t x, uint32_t seed) { return reverse_bits32(laine_karras(reverse_bits32(x), seed)); }
FH_HD uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }

// XOR of the generator-matrix columns selected by the bits of a 32-bit index.
// `row` points at the 52-entry row of the wanted dimension (global memory or LDS).
template <typename Ptr>
FH_HD uint32_t sobol_row(Ptr row, uint32_t index)
{
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) r ^= ((index >> i) & 1u) ? row[i] : 0u;
  return r;
}

// the same XOR from the byte-indexed form of the row ([4][256]: one table per byte of the index, fh_ctx_create): four reads, three XORs
template <typename Ptr>
FH_HD uint32_t sobol_row_bytes(Ptr t, uint32_t index)
{
  return t[index & 255u] ^ t[256u + ((index >> 8) & 255u)] ^ t[512u + ((index >> 16) & 255u)] ^ t[768u + (index >> 24)];
}

// 1-D draw: Owen-scrambled Sobol' point `sobol_index32` in dimension `dim` (sobol.cu:10733-10742).
// sobol_index32 = uint32(image_idx + n_spp*W*H) as seeded by pt.cu:386.
template <typename Ptr>
FH_HD float sobol_draw(Ptr row_of_dim, uint32_t sobol_index32, uint32_t dim, uint32_t seed_hash)
{
  const uint32_t index = owen_scramble(sobol_index32, seed_hash);
  const uint32_t v = owen_scramble(sobol_row(row_of_dim, index), hash_combine(seed_hash, dim));
  return v * (1.0f / 4294967296.0f);
}
template <typename Ptr>
FH_HD float sobol_draw_bytes(Ptr tables_of_dim, uint32_t sobol_index32, uint32_t dim, uint32_t seed_hash)
{
  const uint32_t index = owen_scramble(sobol_index32, seed_hash);
  const uint32_t v = owen_scramble(sobol_row_bytes(tables_of_dim, index), hash_combine(seed_hash, dim));
  return v * (1.0f / 4294967296.0f);
}

// ---- warps (sampling.cu:54-110)
FH_HD f2 concentric_disk(f2 u)
{
  const f2 u0 = 2.0f * u - 1.0f;
  if (u0.x == 0.0f && u0.y == 0.0f) return mk2(0.0f, 0.0f);
  const bool xdom = fabsf(u0.x) > fabsf(u0.y);
  const float r = xdom ? u0.x : u0.y;
  const float theta = xdom ? 0.25f * kPi * u0.y / u0.x : 0.5f * kPi - 0.25f * kPi * u0.x / u0.y;
  float s, c;
  fhe_sincos(theta, &s, &c);
  return mk2(r * c, r * s);
}
FH_HD f3 cosine_hemisphere(f2 u)
{
  const f2 d = concentric_disk(u);
  return mk3(d.x, sqrt_cr(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y)), d.y);
}
FH_HD f2 triangle_barycentric(f2 u)
{
  const float su0 = sqrt_cr(u.x);
  return mk2(1.0f - su0, u.y * su0);
}
// Heitz 2018 visible-normal sampling; phi is formed in double and rounded (the reference mixes M_PI in)
FH_HD f3 sample_vndf(f3 wo, float ax, float ay, f2 u)
{
  const f3 Vh = normalize(mk3(ax * wo.x, wo.y, ay * wo.z));
  const float lensq = Vh.x * Vh.x + Vh.z * Vh.z;
  const f3 T1 = lensq > 0 ? mk3(Vh.z, 0, -Vh.x) / sqrt_cr(lensq) : mk3(0, 0, 1);
  const f3 T2 = cross(Vh, T1);
  const float r = sqrt_cr(u.x);
  const float phi = (float)(2.0f * 3.14159265358979323846 * u.y);
  float sp, cp;
  fhe_sincos(phi, &sp, &cp);
  const float t1 = r * cp;
  float t2 = r * sp;
  const float s = 0.5f * (1.0f + Vh.y);
  t2 = (1.0f - s) * sqrt_cr(fmaxf(1.0f - t1 * t1, 0.0f)) + s * t2;
  const f3 Nh = t1 * T1 + t2 * T2 + sqrt_cr(fmaxf(1.0f - t1 * t1 - t2 * t2, 0.0f)) * Vh;
  return normalize(mk3(ax * Nh.x, fmaxf(0.0f, Nh.y), ay * Nh.z));
}

}  // namespace fh
