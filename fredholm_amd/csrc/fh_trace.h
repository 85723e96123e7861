// fh_trace.h -- ray / BVH traversal and ray / triangle intersection (device code).
//
// Replaces what the reference delegates to the closed-source OptiX runtime: optixTrace for the
// three ray classes (fredholm/modules/pt.cu:82-123) over the GAS/IAS built in
// fredholm/include/fredholm/renderer.h:434-552.  Semantics kept: tmin = 0, no culling, closest hit
// for RADIANCE/LIGHT rays, first hit terminates SHADOW rays, hit attributes (u,v) weight v1,v2.
//
// Triangle test: watertight test of Woop, Benthin, Wald (JCGT 2013) -- OptiX's built-in triangle
// test is watertight too.  Equal-t ties resolve to the lowest face id, which makes the closest
// hit independent of BVH shape and traversal order (needed for seam-free multi-GPU tiling and for
// exact comparison with the CPU checker, which uses a different BVH).  Box tests are conservative
// (boxes padded at build time, far distance inflated), so no accepted triangle is ever culled.
#pragma once
#include "fh_device.h"

namespace fh {

struct RayPre {
  f3 o;        // origin
  f3 inv;      // 1 / direction (zero components replaced by +-1e-20)
  float Sx, Sy, Sz;
  int kx, ky, kz;
};

FH_D float comp(f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// 1 / d with zero components replaced by +-1e-20
FH_D f3 safe_reciprocal(f3 d)
{
  return mk3(1.0f / (fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x), 1.0f / (fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y),
             1.0f / (fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z));
}

FH_D RayPre ray_prepare(f3 o, f3 d)
{
  RayPre r;
  r.o = o;
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  r.kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
  r.kx = r.kz + 1; if (r.kx == 3) r.kx = 0;
  r.ky = r.kx + 1; if (r.ky == 3) r.ky = 0;
  if (comp(d, r.kz) < 0.0f) { const int t = r.kx; r.kx = r.ky; r.ky = t; }
  const float dz = comp(d, r.kz);
  r.Sx = comp(d, r.kx) / dz;
  r.Sy = comp(d, r.ky) / dz;
  r.Sz = 1.0f / dz;
  r.inv = safe_reciprocal(d);
  return r;
}

// returns true with t >= 0 and barycentrics (bu, bv) of v1, v2
FH_D bool tri_test(const RayPre& r, f3 p0, f3 p1, f3 p2, float& t, float& bu, float& bv)
{
  const f3 A = p0 - r.o, B = p1 - r.o, C = p2 - r.o;
  const float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
  const float Ax = fmaf(-r.Sx, Akz, comp(A, r.kx)), Ay = fmaf(-r.Sy, Akz, comp(A, r.ky));
  const float Bx = fmaf(-r.Sx, Bkz, comp(B, r.kx)), By = fmaf(-r.Sy, Bkz, comp(B, r.ky));
  const float Cx = fmaf(-r.Sx, Ckz, comp(C, r.kx)), Cy = fmaf(-r.Sy, Ckz, comp(C, r.ky));
  float U = Cx * By - Cy * Bx;
  float V = Ax * Cy - Ay * Cx;
  float W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {  // edge-on: redo the edge functions in fp64
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = U + V + W;
  if (det == 0.0f) return false;
  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = fmaf(W, Cz, fmaf(V, Bz, U * Az));
  const float rcp = 1.0f / det;
  t = T * rcp;
  if (!(t >= 0.0f)) return false;
  t = fabsf(t);  // -0.0 -> +0.0: hit distances order like their bit patterns
  bu = V * rcp;
  bv = W * rcp;
  return true;
}

struct HitRec { float t, u, v; uint32_t prim; };

// conservative slab test against (lo, hi); returns entry distance in tn.  Every plane distance is moved by 2^-21 of itself to the safe side (entries earlier,
// exits later): (p - o) * inv carries a rounding error of 2^-24 of |p - o| per axis, which for a ray that starts far outside the box is more than the box's padding
FH_D bool slab_test(const RayPre& r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tmax, float& tn)
{
  const float k = 4.76837158203125e-7f;
  float t0 = (lox - r.o.x) * r.inv.x, t1 = (hix - r.o.x) * r.inv.x;
  float n = fminf(t0, t1), f = fmaxf(t0, t1);
  float tnr = fmaf(fabsf(n), -k, n), tf = fmaf(fabsf(f), k, f);
  t0 = (loy - r.o.y) * r.inv.y; t1 = (hiy - r.o.y) * r.inv.y;
  n = fminf(t0, t1); f = fmaxf(t0, t1);
  tnr = fmaxf(tnr, fmaf(fabsf(n), -k, n)); tf = fminf(tf, fmaf(fabsf(f), k, f));
  t0 = (loz - r.o.z) * r.inv.z; t1 = (hiz - r.o.z) * r.inv.z;
  n = fminf(t0, t1); f = fmaxf(t0, t1);
  tnr = fmaxf(tnr, fmaf(fabsf(n), -k, n)); tf = fminf(tf, fmaf(fabsf(f), k, f));
  tn = tnr;
  return tnr <= tf && tf >= 0.0f && tnr <= tmax;
}

// __anyhit__{radiance,shadow,light} (pt.cu:545-678): a candidate hit is ignored when the base-colour texture's alpha or
// the alpha texture's red channel is below 0.5 at the hit's texture coordinate
FH_D f3 tex_rgb(const SceneDev& sc, int id, float u, float v)
{
  float o[4];
  fht_tex2d(&sc.textures[id], sc.srgb_lut, u, v, o);
  return mk3(o[0], o[1], o[2]);
}
FH_D float4 tex_rgba(const SceneDev& sc, int id, float u, float v)
{
  float o[4];
  fht_tex2d(&sc.textures[id], sc.srgb_lut, u, v, o);
  return make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __attribute__((noinline)) bool alpha_pass(const SceneDev& sc, uint32_t prim, float bu, float bv)
{
  // one 64-byte record per face (capi.hip: rebuild_device_scene): texture coordinates of the three vertices + the textures that can actually cut
  // (a texture whose every texel is opaque is not listed: a filtered fetch of it cannot come out below 0.5)
  const uint4* r = sc.alpha_rec + 8 * (size_t)prim;
  {  // the face's opacity micromap (capi.hip: rebuild_device_scene): 16 x 16 cells of the hit's barycentrics, two bits each -- decided at upload for every point of the cell
    const uint32_t ci = (uint32_t)fminf(fmaxf(bu, 0.0f) * 16.0f, 15.0f), cj = (uint32_t)fminf(fmaxf(bv, 0.0f) * 16.0f, 15.0f);
    const uint32_t cell = cj * 16u + ci;
    const uint32_t st = (((const uint32_t*)(r + 4))[cell >> 4] >> (2u * (cell & 15u))) & 3u;
    if (st == 1u) return true;
    if (st == 2u) return false;
  }
  const uint4 q0 = r[0], q1 = r[1], q2 = r[2];  // (the alpha texture's entry, r[3], is read where a face has one: most cut-outs sit in the base colour's alpha)
  const float bw = 1.0f - bu - bv;
  const float tu = bw * __uint_as_float(q0.x) + bu * __uint_as_float(q0.z) + bv * __uint_as_float(q1.x);
  const float tv = bw * __uint_as_float(q0.y) + bu * __uint_as_float(q0.w) + bv * __uint_as_float(q1.y);
  const uint32_t flags = q1.z;
  if (flags & 1u) {  // alpha of the base-colour texture
    const uint8_t* tex = (const uint8_t*)(const __attribute__((address_space(1))) uint8_t*)(uintptr_t)(((unsigned long long)q2.y << 32) | q2.x);  // (a pointer into global memory: four global loads instead of flat ones)
    if (fht_tex2d_channel8(tex, q2.z, q2.w, nullptr, 3u, tu, tv) < 0.5f) return false;
  }
  if (flags & 2u) {  // red of the alpha texture
    const uint4 q3 = r[3];
    const uint8_t* tex = (const uint8_t*)(const __attribute__((address_space(1))) uint8_t*)(uintptr_t)(((unsigned long long)q3.y << 32) | q3.x);
    if (fht_tex2d_channel8(tex, q3.z, q3.w, (flags & 4u) ? sc.srgb_lut : nullptr, 0u, tu, tv) < 0.5f) return false;
  }
  return true;
}

// accept a candidate under the closest-hit order (t, then face id)
FH_D bool closer(float t, uint32_t prim, const HitRec& best) { return t < best.t || (t == best.t && prim < best.prim); }

// ---------------------------------------------------------------------------------------------
// BVH2 (one 64-byte node = both child boxes + both child references).  Child reference >= 0:
// inner node index; < 0: leaf, ~ref = (first_triangle << 3) | (count - 1).
// ---------------------------------------------------------------------------------------------
constexpr int kBvh2Stack = 96;

template <bool ANY_HIT, bool COUNT, bool ALPHA = false>
FH_D bool traverse_bvh2(const Bvh2Dev& bvh, f3 o, f3 d, float tmax, HitRec& best, uint32_t& n_nodes, uint32_t& n_tris, const SceneDev* sc = nullptr)
{
  best.t = tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xffffffffu;
  if (bvh.n_nodes == 0) return false;
  const RayPre r = ray_prepare(o, d);
  int stack[kBvh2Stack];
  int sp = 0;
  int cur = 0;
  bool found = false;
  for (;;) {
    if (cur >= 0) {
      const float4 n0 = bvh.nodes[4 * cur], n1 = bvh.nodes[4 * cur + 1], n2 = bvh.nodes[4 * cur + 2], n3 = bvh.nodes[4 * cur + 3];
      if (COUNT) n_nodes++;
      float ta, tb;
      const bool ha = slab_test(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, best.t, ta);
      const bool hb = slab_test(r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, best.t, tb);
      const int ca = __float_as_int(n3.x), cb = __float_as_int(n3.y);
      if (ha && hb) {
        const bool a_first = ta <= tb;
        cur = a_first ? ca : cb;
        if (sp < kBvh2Stack) stack[sp++] = a_first ? cb : ca;
      } else if (ha) cur = ca;
      else if (hb) cur = cb;
      else { if (sp == 0) break; cur = stack[--sp]; }
    } else {
      const uint32_t ref = (uint32_t)~cur;
      const uint32_t first = ref >> 3, count = (ref & 7u) + 1u;
      for (uint32_t i = first; i < first + count; ++i) {
        const float4 a = bvh.tris[3 * i], b = bvh.tris[3 * i + 1], c = bvh.tris[3 * i + 2];
        if (COUNT) n_tris++;
        float t, bu, bv;
        if (!tri_test(r, mk3(a), mk3(b), mk3(c), t, bu, bv)) continue;
        if (t > tmax) continue;
        const uint32_t prim = __float_as_uint(a.w);
        if (found && !closer(t, prim, best)) continue;
        if (ALPHA && b.w != 0.0f && !alpha_pass(*sc, prim, bu, bv)) continue;
        best.t = t; best.u = bu; best.v = bv; best.prim = prim;
        found = true;
        if (ANY_HIT) return true;
      }
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
  return found;
}


// ---------------------------------------------------------------------------------------------
// BVH8: 64-byte nodes with eight children whose boxes are quantised to 8 bits per plane relative
// to the node's origin and per-axis power-of-two scale (after Ylitie, Karras, Laine, "Efficient
// Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017, with everything that is
// not a plane squeezed into the first 16 bytes so that a node is FOUR 16-byte loads and never straddles a 128-byte line):
//   n0 = origin.x | ex, origin.y | ey, origin.z | ez, first inner-child node index << 8 | imask
//        The biased exponent of an axis' scale sits in the low mantissa byte of that axis' origin; the origin is the float
//        the word spells WITH that byte (the builder picks it at or below the node's lower corner and quantises against it).
//   n1 = qlo_x[0..7], qlo_y[0..7]    n2 = qlo_z[0..7], qhi_x[0..7]    n3 = qhi_y[0..7], qhi_z[0..7]
// Child slot i is an inner node when bit i of imask is set (the inner children of a node are consecutive nodes, in slot
// order), else a leaf holding the ONE triangle in slot 8 * node + i of the triangle array, else empty: an empty slot has the
// inverted box (lo 255, hi 0) no ray enters, and its triangle slot holds a degenerate triangle no ray hits.
// A node test yields one bit per child slot.  Inner hits are then permuted by the ray's octant (bit i -> bit i ^ oct) so that
// "highest bit first" visits the children front to back.
// ---------------------------------------------------------------------------------------------
constexpr int kBvh8Stack = 48;
constexpr uint32_t kBvh8NodeVec = 4;  // uint4 per node

struct Ray8 {
  f3 o, inv;       // origin, safe reciprocal direction
  uint32_t oct;    // octant: bit 2 / 1 / 0 set when the x / y / z direction is positive
  bool nx, ny, nz; // direction signs
};

// bit i of an 8-bit mask -> bit i ^ oct
FH_D uint32_t octant_permute(uint32_t m, uint32_t oct)
{
  const uint32_t s4 = oct & 4u, s2 = oct & 2u, s1 = oct & 1u;
  m = ((m * 0x0101u) >> s4) & 0xffu;                          // oct & 4: swap the nibbles
  m = ((m << s2) & 0xccu) | ((m >> s2) & 0x33u);              // oct & 2: swap the bit pairs (a shift of 0 leaves m as it is)
  m = ((m << s1) & 0xaau) | ((m >> s1) & 0x55u);              // oct & 1: swap neighbours
  return m;
}

// The node test in units of the ray's current limit.  All distances are computed as t' = t / tmax: the planes' conversions and the
// min / max / compare instructions are what a node costs (gfx950 issues v_fma / v_mul / v_add_f32 in ~2.2 cycles per wave BESIDE the ~4.1-cycle
// instructions of everything else, profiles/r03_issue_peak.txt, and a node has ~2 of the second kind for each of the first), so the test is
// arranged to need as few of those as possible:
//   - the clamp of a v_fma_f32 (a free output modifier) bounds every plane distance to [0, 1] = [ray start, tmax]: no max(., 0), no min(., tmax);
//   - with both ends inside [0, 1] the ray enters the box iff max3(near) < min3(far); the difference is one v_sub_f32 (the fast class) whose sign
//     bit a v_alignbit_b32 shifts into the mask: one slow instruction where compare + add-with-carry were two.
// Per child: 6 conversions, max3, min3, alignbit = 9 of the slow kind (12 before), plus v_min + v_rcp per node for 1 / tmax.
// The comparison is strict because both sides saturate: a box behind the ray gives 0 < 0, one beyond tmax 1 < 1, an empty slot (lo 255, hi 0) never has
// near < far.  A box the ray really enters has near' < far' by what the build's padding and the slack below put between them, and 1 / tmax is rounded
// down by two units (v_rcp_f32 is good to one), so a box that starts just before tmax is never cut off.  tmax < 0: every scaled interval is reversed, so no child with a proper box is hit, as with the unscaled
// comparison (an empty slot's inverted box can be flagged then: its triangle slot holds the degenerate triangle no ray hits); tmax = 0: the scale is infinite and children may be flagged, but no triangle is accepted (the ray's hit record starts at its tmax); a NaN tmax
// counts as no limit in v_min_f32, as it did in the minimum of the unscaled form.  (tmax = +0 no longer reaches the division: see the floor in node8_test.)
FH_D float fma_clamp01(float a, float b, float c)
{
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <int J>
FH_D void node8_child(uint32_t& hits, uint32_t nearx, uint32_t farx, uint32_t neary, uint32_t fary, uint32_t nearz, uint32_t farz, float sx, float sy, float sz, float nx, float ny, float nz,
                      float fx, float fy, float fz)
{
  const float t0x = fma_clamp01((float)((nearx >> (8 * J)) & 0xffu), sx, nx), t1x = fma_clamp01((float)((farx >> (8 * J)) & 0xffu), sx, fx);
  const float t0y = fma_clamp01((float)((neary >> (8 * J)) & 0xffu), sy, ny), t1y = fma_clamp01((float)((fary >> (8 * J)) & 0xffu), sy, fy);
  const float t0z = fma_clamp01((float)((nearz >> (8 * J)) & 0xffu), sz, nz), t1z = fma_clamp01((float)((farz >> (8 * J)) & 0xffu), sz, fz);
  float tn, tf;  // (as instructions: the operands are clamped, never NaN, and need no canonicalising first)
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(t0x), "v"(t0y), "v"(t0z));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(t1x), "v"(t1y), "v"(t1z));
  hits = __builtin_amdgcn_alignbit(hits, __float_as_uint(tn - tf), 31u);  // hits = 2 * hits + (tn < tf)
}

// one bit per child slot whose (conservative) box the ray enters before tmax
FH_D uint32_t node8_test(const Ray8& r, const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, float tmax)
{
  // The unit: tmax kept inside [1e-12, 1e30].  The upper end gives a ray without a limit a finite unit.  The lower end is for a ray whose best hit so far is AT its origin
  // (t = +0: an origin on a triangle): another triangle at t = 0 with a lower face id still has to be found -- the closest-hit order breaks ties by face id, whatever the
  // shape of the tree -- but 1 / 0 is infinite and an infinite scale turns plane distances into NaN; with the floor every box that holds the origin is entered (near 0,
  // far 1) and every box that starts later is not.  Taken as an UNSIGNED maximum of the bit patterns: for tmax >= +0 that is the float maximum, a negative tmax (a shadow
  // ray shorter than its epsilon) and NaN keep their bits, so the one still flags nothing and the other still counts as no limit.
  float tl;
  asm("v_max_u32 %0, 0x2b8cbccc, %1\n\tv_min_f32 %0, 0x7149f2ca, %0" : "=v"(tl) : "v"(tmax));
  const float rt = __builtin_amdgcn_rcpf(tl) * 0.99999976158142090f;
  const float ix = r.inv.x * rt, iy = r.inv.y * rt, iz = r.inv.z * rt;
  const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
  // scale of an axis = 2^(e - 127) with e the low byte of the origin word (<< 23 shifts everything else out; bit 8 lands in the sign, hence the fabsf, a free
  // source modifier)
  const float kx = fabsf(__uint_as_float(n0.x << 23)), ky = fabsf(__uint_as_float(n0.y << 23)), kz = fabsf(__uint_as_float(n0.z << 23));
  const float sx = kx * ix, sy = ky * iy, sz = kz * iz;
  const float ox = (px - r.o.x) * ix, oy = (py - r.o.y) * iy, oz = (pz - r.o.z) * iz;
  // The child boxes carry the build's absolute padding (2^-16 of the scene's largest coordinate), ~100 x the rounding error of these distances while the ray
  // starts within a few hundred scene sizes of the node.  The error of (p - o) * inv grows with |p - o| though (2^-24 of it, per axis), and so does what the
  // triangle test itself makes of a ray from far away; so the near planes of every axis are moved in and the far planes out by 2^-21 of that axis' own offset:
  // nothing next to the padding for a ray that starts in or near the scene, and what keeps a thin box from being skipped by a camera far outside it.
  // Six FMA-class instructions per node, which issue beside the others.
  const float kSlack = 4.76837158203125e-7f;
  const float fx = fmaf(fabsf(ox), kSlack, ox), fy = fmaf(fabsf(oy), kSlack, oy), fz = fmaf(fabsf(oz), kSlack, oz);
  const float nx = fmaf(fabsf(ox), -kSlack, ox), ny = fmaf(fabsf(oy), -kSlack, oy), nz = fmaf(fabsf(oz), -kSlack, oz);
  uint32_t hits = 0;
  {  // slots 7 .. 4, then 3 .. 0: every test shifts the mask left and moves its result in at the bottom
    const uint32_t nearx = r.nx ? n2.w : n1.y, farx = r.nx ? n1.y : n2.w, neary = r.ny ? n3.y : n1.w, fary = r.ny ? n1.w : n3.y, nearz = r.nz ? n3.w : n2.y, farz = r.nz ? n2.y : n3.w;
    node8_child<3>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<2>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<1>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<0>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
  }
  {
    const uint32_t nearx = r.nx ? n2.z : n1.x, farx = r.nx ? n1.x : n2.z, neary = r.ny ? n3.x : n1.z, fary = r.ny ? n1.z : n3.x, nearz = r.nz ? n3.z : n2.x, farz = r.nz ? n2.x : n3.z;
    node8_child<3>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<2>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<1>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
    node8_child<0>(hits, nearx, farx, neary, fary, nearz, farz, sx, sy, sz, nx, ny, nz, fx, fy, fz);
  }
  return hits;
}

FH_D Ray8 ray8_prepare(const RayPre& rp, f3 d)
{
  Ray8 r;
  r.o = rp.o;
  r.inv = rp.inv;
  (void)d;
  r.nx = r.inv.x < 0.0f; r.ny = r.inv.y < 0.0f; r.nz = r.inv.z < 0.0f;  // sign of the reciprocal actually used (-0.0 components count as negative)
  r.oct = (r.nx ? 0u : 4u) | (r.ny ? 0u : 2u) | (r.nz ? 0u : 1u);
  return r;
}

// (A node fetch by PAIRS of lanes -- the two lanes of a pair read both their nodes together, 32 contiguous bytes each, one butterfly of DPP selects hands every lane its halves:
// half the L1 look-ups per visit -- was built and measured in round 4 and is slower: the transposition costs more than the look-ups.  tools/patches/r6_pruned_switches.patch)
// one visited node: the group of its inner children the ray enters (first-child node index, octant-ordered hit bits << 24 | imask)
// and the group of its candidate triangles (first triangle slot, one bit per slot)
FH_D void node8_eval(const Ray8& r, uint32_t ni, const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, float tmax, uint2& group, uint2& tg)
{
  const uint32_t hm = node8_test(r, n0, n1, n2, n3, tmax);
  const uint32_t imask = n0.w & 0xffu;
  group = make_uint2(n0.w >> 8, (octant_permute(hm & imask, r.oct) << 24) | imask);
  tg = make_uint2(8u * ni, hm & ~imask);
}
// The top of the tree in LDS (streaming kernels).  Every ray visits the root and one or two of its children: 2.5 of a secondary ray's 15 node visits on the
// soup, 17 % of the node fetches -- each of them four vector-L1 look-ups per lane, and the look-up rate of the L1 (one line per cycle and CU) is one of the two limits
// the kernels sit on.  A workgroup stages nodes 0 .. 8 (the root and its inner children, which the breadth-first collapse numbers 1 ..) once, 576 bytes,
// and a visit of one of them reads LDS (8 cycles per 64-lane ds_read_b128) instead.  Secondary and merged launches only; alone on the GPU the secondary launch takes
// 103.3 -> 100.7 ms per configs[2] frame and 783 -> 743 ms per 512 spp of configs[3] (profiles/README.md r5-7).
constexpr uint32_t kTopNodes = 9;
constexpr uint32_t kTopLdsBytes = kTopNodes * 64u;
FH_D void stage_top_nodes(const Bvh8Dev& bvh, uint4* lds_top)  // every thread of the workgroup; ends with its barrier
{
  if (threadIdx.x < kTopNodes * 4u) lds_top[threadIdx.x] = (threadIdx.x >> 2) < bvh.n_nodes ? bvh.nodes[threadIdx.x] : make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
}
template <bool ORDERED = true>
FH_D void node8_visit(const Bvh8Dev& bvh, const Ray8& r, uint32_t ni, float tmax, uint2& group, uint2& tg, const uint4* top = nullptr)
{
  uint4 n0, n1, n2, n3;
  if (top && ni < kTopNodes) {
    const uint4* nd = top + 4u * ni;
    n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3];
  } else {
    // (a 32-bit byte offset -- the builder refuses trees of 2^23 nodes -- lets the four loads share the base in scalar registers)
    const uint4* nd = (const uint4*)((const char*)bvh.nodes + (ni << 6));
    n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3];
  }
  const uint32_t hm = node8_test(r, n0, n1, n2, n3, tmax);
  const uint32_t imask = n0.w & 0xffu;
  if (!ORDERED) {
    group = make_uint2(n0.w >> 8, ((hm & imask) << 24) | imask);
    tg = make_uint2(8u * ni, hm & ~imask);
    return;
  }
  // (the permutation is 15 instructions; reading it from a 2 KB table in global memory instead -- one instruction and a byte load -- made the streaming kernels
  // 4-7 % SLOWER, configs[2] closest 71.4 -> 74.7 ms, secondary 111 -> 118.5 ms: a fifth load per node visit costs more in the memory pipeline than 14 instructions
  // cost in the VALU; LDS has no room for the table without giving up a workgroup per CU)
  group = make_uint2(n0.w >> 8, (octant_permute(hm & imask, r.oct) << 24) | imask);
  tg = make_uint2(8u * ni, hm & ~imask);
}

// the three float4 of triangle slot `slot` (48 bytes each).  The byte offset fits 32 bits (kCoopMaxTris slots) and is built from shifts and adds:
// a 32-bit multiply is a quarter-rate instruction, a 64-bit address two more
FH_D const float4* tri_slot_ptr(const Bvh8Dev& bvh, uint32_t slot)
{
  uint32_t s3;
  asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(s3) : "v"(slot));
  return (const float4*)((const char*)bvh.tris + (s3 << 4));
}

// wave-level step counters of the instrumented build: how many times a wave executed the node test /
// the triangle test, whatever the number of lanes taking part (SIMD efficiency = lane steps / (64 * wave steps))
struct WaveSteps { uint32_t node = 0, tri = 0; };
FH_D bool first_active_lane() { return __lane_id() == (uint32_t)__ffsll((long long)__ballot(true)) - 1u; }

// Traversal stack of node groups.  A group is pushed only while descending into one of its children, so the stack never holds
// more entries than the tree has levels; the builder records that number (fh_ctx::bvh8_depth) and refuses trees deeper than kBvh8Stack.
//   GroupStack<true>   every entry in LDS, column layout [entry][thread] (conflict-free; constructed from the block's LDS and the tree depth); the launcher sizes the dynamic LDS for the
//                      tree's depth (the streaming kernels may spill the deepest levels to global memory, below), so there is no private array: `sp` lives in a register.  (The first build kept
//                      6 entries in LDS and the rest in a private array; the compiler then kept the whole object, `sp` included, in scratch:
//                      every push and pop started with a scratch load on the dependent path -- 408 B of scratch, ~1 GB of writes per launch.)
//   GroupStack<false>  a private array, for the per-lane loops of k_tail and the batch queries.
template <bool LDS>
struct GroupStack;
template <>
struct GroupStack<false> {
  int sp = 0;
  uint2 spill[kBvh8Stack];
  FH_D GroupStack(uint2*, int) {}
  FH_D GroupStack(uint2*, int, uint2*, uint32_t, uint32_t) {}
  FH_D void push(uint2 g) { if (sp < kBvh8Stack) spill[sp++] = g; }
  FH_D uint2 pop() { return spill[--sp]; }
  FH_D void set_anchor(uint32_t node, uint32_t skip) { spill[0] = make_uint2(node, skip); }
  FH_D void set_anchor_word(uint32_t v) { spill[0].x = v; }
  FH_D uint32_t anchor() const { return spill[0].x; }
  FH_D uint32_t anchor_skip() const { return spill[0].y; }
};
template <>
struct GroupStack<true> {
  // 5 bytes per entry: a word (first-child node index << 8 | pending hit bits) and a byte (the group's inner-child mask), each in its own
  // column array [entry][thread] -- 1.25 KB per tree level and workgroup instead of 2 KB, which is a workgroup more per CU on every tree
  // deeper than eight levels (node indices stay below 2^24: the builder refuses larger trees)
  // The streaming kernels may keep only the first `cap` levels there (StackSpill): deeper entries go to a column of global memory, [entry][thread of the launch]
  // like the LDS part, so that a tree of many levels -- whose deep entries are rarely reached -- does not cost a workgroup per CU.
  uint32_t* word;
  uint8_t* mask;
  int sp = 0;
  int cap = kBvh8Stack;
  uint2* over = nullptr;   // the spill area of the launch (wave-uniform: it stays in scalar registers; the thread's column is formed where an entry is spilled)
  uint32_t ostride = 0;    // threads of the launch
  FH_D GroupStack(uint2* block_lds, int depth) : word((uint32_t*)block_lds + threadIdx.x), mask((uint8_t*)((uint32_t*)block_lds + depth * 256) + threadIdx.x) {}
  FH_D GroupStack(uint2* block_lds, int depth, uint2* spill, uint32_t threads, uint32_t)
      : word((uint32_t*)block_lds + threadIdx.x), mask((uint8_t*)((uint32_t*)block_lds + depth * 256) + threadIdx.x), cap(spill ? depth : kBvh8Stack), over(spill), ostride(threads) {}
  FH_D uint2* over_at(int level) const
  {
    uint32_t column = blockIdx.x * blockDim.x + threadIdx.x;
    asm volatile("" : "+v"(column));  // (formed HERE: hoisted out of the traversal loop, the thread's column pointer was a 64-bit value the any-hit kernels kept in 8 bytes of scratch)
    return over + ((size_t)(uint32_t)(level - cap) * ostride + column);
  }
  FH_D void push(uint2 g)
  {
    if (sp < cap) { word[sp * 256] = (g.x << 8) | (g.y >> 24); mask[sp * 256] = (uint8_t)g.y; }
    else *over_at(sp) = g;
    ++sp;
  }
  FH_D uint2 pop()
  {
    --sp;
    if (sp >= cap) return *over_at(sp);
    const uint32_t w = word[sp * 256];
    return make_uint2(w >> 8, (w << 24) | mask[sp * 256]);
  }
  // entry 0 of a ray that started below the root (traverse_stream: bottom-up start) is not a group: it holds the node the ray climbs from next and, while that node's
  // parent is being visited, the child slot of the parent the ray came up through (always in LDS: every configuration keeps at least one level there)
  FH_D void set_anchor(uint32_t node, uint32_t skip) { word[0] = node; mask[0] = (uint8_t)skip; }
  FH_D void set_anchor_word(uint32_t v) { word[0] = v; }
  FH_D uint32_t anchor() const { return word[0]; }
  FH_D uint32_t anchor_skip() const { return mask[0]; }
};
// where the streaming kernels put stack entries beyond the LDS part (null: everything in LDS)
struct StackSpill { uint2* area; uint32_t lds_entries; uint32_t probe = 0; };  // lds_entries: levels the launch keeps in LDS; probe: the launch adds its test rounds to the bounce's
                                                                              // CNT_COST_* words (two atomics per wave: only the passes that decide where rays start ask for it, render.hip)
// dynamic LDS of one 256-thread workgroup whose lanes keep `depth` stack entries there
FH_HD uint32_t lds_stack_bytes(uint32_t depth) { return (depth * 256u * 5u + 15u) & ~15u; }
// Entries a traversal stack needs for a tree of `levels` node levels: a group is pushed while the ray descends into one of its nodes with siblings still to
// visit; the root's group holds the root alone and is never pushed, so the groups that can be on the stack at once are those of levels 1 .. levels - 1.
// (One entry fewer than levels is 1.25 KB per workgroup: five workgroups per CU instead of four on the 15-level tree of the Sponza-class scene with the whole stack in LDS.)
FH_HD uint32_t stack_entries_for(uint32_t levels) { return levels < 2u ? 1u : levels - 1u; }

template <bool ANY_HIT, bool COUNT, bool LDS = false, bool ALPHA = false>
FH_D bool traverse_bvh8(const Bvh8Dev& bvh, f3 o, f3 d, float tmax, HitRec& best, uint32_t& n_nodes, uint32_t& n_tris, WaveSteps* ws = nullptr, uint2* lds_column = nullptr,
                        int lds_stride = 0, const SceneDev* sc = nullptr)  // (LDS: lds_column = the block's stack area, lds_stride = tree depth)
{
  best.t = tmax; best.u = 0.0f; best.v = 0.0f; best.prim = 0xffffffffu;
  if (bvh.n_nodes == 0) return false;
  const RayPre rp = ray_prepare(o, d);
  const Ray8 r = ray8_prepare(rp, d);
  GroupStack<LDS> stack(lds_column, lds_stride);
  uint2 group = make_uint2(0u, 0x80000000u);
  bool found = false;
  for (;;) {
    uint2 tg = make_uint2(0u, 0u);
    if (group.y & 0xff000000u) {
      const uint32_t hits_imask = group.y;
      const uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
      group.y &= ~(1u << bit);
      if (group.y & 0xff000000u) stack.push(group);
      const uint32_t slot = (bit - 24u) ^ r.oct;
      const uint32_t rel = (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
      const uint32_t ni = group.x + rel;
      if (COUNT) { n_nodes++; if (ws && first_active_lane()) ws->node++; }
      node8_visit(bvh, r, ni, best.t, group, tg);
    } else {
      tg = group;
      group = make_uint2(0u, 0u);
    }
    while (tg.y) {
      const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
      tg.y &= tg.y - 1u;
      const float4* tp = tri_slot_ptr(bvh, tg.x + b);
      const float4 a = tp[0], bb = tp[1], c = tp[2];
      if (COUNT) { n_tris++; if (ws && first_active_lane()) ws->tri++; }
      float t, bu, bv;
      if (!tri_test(rp, mk3(a), mk3(bb), mk3(c), t, bu, bv)) continue;
      if (t > tmax) continue;
      const uint32_t prim = __float_as_uint(a.w);
      if (found && !closer(t, prim, best)) continue;
      if (ALPHA && bb.w != 0.0f && !alpha_pass(*sc, prim, bu, bv)) continue;
      best.t = t; best.u = bu; best.v = bv; best.prim = prim;
      found = true;
      if (ANY_HIT) return true;
    }
    if ((group.y & 0xff000000u) == 0u) {
      if (stack.sp == 0) break;
      group = stack.pop();
    }
  }
  return found;
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative BVH8 traversal.  In the per-lane loop above a triangle test runs with 3-6 of 64 lanes
// active (a node produces ~0.3 candidate triangles per lane), so more than half of a wave's issue slots
// go to nearly empty triangle tests.  Here candidate triangles are not tested by the lane that found
// them: they go into a wave-private LDS queue as (owner lane, triangle), and once the queue holds
// `flush` entries ALL lanes take one entry each, fetch the owner's ray from LDS and test it.  The closest
// hit of every ray lives in LDS as one 64-bit key (t bits << 32 | face id): the closest-hit order
// "t, then lowest face id" is the unsigned order of that key for t >= 0, so concurrent testers commit with
// ds_min_u64 and the lane whose key survives also stores the barycentrics.  The result is the minimum over
// the same candidate set as the per-lane loop, i.e. bit-identical.  Must be called by all 64 lanes of a
// wave together (lanes without a ray pass valid = false).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kCoopQueue = 128;      // ring of (triangle << 6 | owner lane); >= 64 + flush threshold
constexpr uint32_t kCoopMaxTris = 1u << 26;
struct CoopLds {            // per-wave slices of the block's LDS
  float4* ray;              // [2][64]: (o.xyz, Sx), (Sy, Sz, kx|ky<<2|kz<<4, -)
  unsigned long long* key;  // [64]
  float2* uv;               // [64]
  uint32_t* queue;          // [kCoopQueue]
  uint4* aring = nullptr;   // [kAlphaRing] candidates waiting for their any-hit test + one counter word behind them (streaming kernels of scenes with cut-outs, alpha_ring)
};
// Pending any-hit tests (scenes with cut-outs, streaming kernels).  A candidate on a face that can cut needs alpha_pass -- a 64-byte record, four texels, ~200 instructions --
// before it may be committed, and inside coop_test a wave runs all of that for the two or three of its 64 lanes whose candidate is such a face, round after round (the Sponza-class
// scene: 520 VALU instructions per wave-level node visit against 250 without cut-outs).  So those lanes only park (owner lane, face, t, u, v) in a per-wave ring and go on; the
// ring is worked off by as many lanes as it holds entries -- when it fills up, once it holds kAlphaFlush of them, and always before a finished ray is committed.  What a ray
// ends up with is the minimum over the same accepted candidates as before (a parked candidate only keeps the ray's limit and its first-hit stop from taking effect a few visits
// earlier): the bits do not change.
constexpr uint32_t kAlphaRing = 32, kAlphaFlush = 16;
// which streaming kernels park: the closest-hit launch.  A ray that stops at its first hit (every secondary ray of a scene without emitters) is finished by the first candidate
// that passes, and parked it goes on walking the tree until the ring is worked off: secondary 760 -> 835 ms per 512 spp of configs[3] with parking, closest 357 -> 329 (r4-19).
// Round 5 built what that suggests -- the owner of a parked candidate SUSPENDED (a bit per lane next to the ring's counter, set by whoever parks, cleared
// by alpha_flush, takes the lane out of the node visits and counts it as idle for the refill trigger) -- and measured it on the same box: secondary 752 ms testing in place,
// 831 parked and walking on, 876 parked and suspended (profiles/README.md r5-3; tools/patches/r6_pruned_switches.patch).  The loss is not the walking: it is the ring's LDS (seven stack levels in LDS become five),
// 32 bytes of scratch instead of 16, and lanes that wait for a round of sixteen.  In place it stays.
template <bool MIXED, bool ALPHA>
struct AlphaDefer { static constexpr bool value = ALPHA && !MIXED; };
constexpr uint32_t kAlphaLdsBytesPerWave = kAlphaRing * 16 + 16;
constexpr uint32_t kAlphaLdsBytesPerBlock = 4u * kAlphaLdsBytesPerWave;
FH_D void alpha_ring(CoopLds& cl, unsigned char* block_lds, uint32_t wave_in_block)
{
  cl.aring = (uint4*)(block_lds + (size_t)wave_in_block * kAlphaLdsBytesPerWave);
  if (__lane_id() == 0u) *(uint32_t*)(cl.aring + kAlphaRing) = 0u;
}
constexpr uint32_t kCoopLdsBytesPerWave = 64 * 32 + 64 * 8 + 64 * 8 + kCoopQueue * 4;
// static LDS of one 256-thread workgroup of a cooperative / streaming traversal kernel (the stack comes on top, dynamically)
constexpr uint32_t kCoopLdsBytesPerBlock = 4u * kCoopLdsBytesPerWave;
FH_D CoopLds coop_lds(unsigned char* block_lds, uint32_t wave_in_block)
{
  unsigned char* b = block_lds + (size_t)wave_in_block * kCoopLdsBytesPerWave;
  CoopLds c;
  c.ray = (float4*)b;
  c.key = (unsigned long long*)(b + 64 * 32);
  c.uv = (float2*)(b + 64 * 32 + 64 * 8);
  c.queue = (uint32_t*)(b + 64 * 32 + 64 * 8 + 64 * 8);
  return c;
}

// one queued candidate: test triangle (e >> 6) against the ray of lane (e & 63), commit into that lane's LDS record
template <bool ANY_HIT, bool COUNT, bool ALPHA, bool DEFER = false>
FH_D void coop_test(const Bvh8Dev& bvh, const CoopLds& cl, uint32_t e, uint32_t& n_tris, WaveSteps* ws, const SceneDev* sc)
{
  const uint32_t owner = e & 63u;
  const float4* tp = tri_slot_ptr(bvh, e >> 6);
  const float4 a = tp[0], bb = tp[1], c = tp[2];
  if (COUNT) { n_tris++; if (ws && first_active_lane()) ws->tri++; }
  const float4 r0 = cl.ray[owner], r1 = cl.ray[64 + owner];
  if (ANY_HIT && r1.w != 0.0f && (uint32_t)cl.key[owner] != 0xffffffffu) return;  // the owner's ray stops at its first hit and has one
  RayPre rp;
  rp.o = mk3(r0.x, r0.y, r0.z);
  rp.Sx = r0.w; rp.Sy = r1.x; rp.Sz = r1.y;
  const uint32_t kk = __float_as_uint(r1.z);
  rp.kx = (int)(kk & 3u); rp.ky = (int)((kk >> 2) & 3u); rp.kz = (int)((kk >> 4) & 3u);
  float t, bu, bv;
  if (!tri_test(rp, mk3(a), mk3(bb), mk3(c), t, bu, bv)) return;
  const uint32_t prim = __float_as_uint(a.w);
  const unsigned long long mine = ((unsigned long long)__float_as_uint(t) << 32) | prim;
  if (mine >= cl.key[owner]) return;  // also rejects t > tmax: the record starts at (tmax, 0xffffffff)
  if (ALPHA && bb.w != 0.0f) {
    if (DEFER) {  // park it (above); a full ring: test in place (face ids stay below 2^26: the builder refuses trees of 2^23 nodes)
      const uint32_t pos = atomicAdd((uint32_t*)(cl.aring + kAlphaRing), 1u);
      if (pos < kAlphaRing) {
        cl.aring[pos] = make_uint4((owner << 26) | prim, __float_as_uint(t), __float_as_uint(bu), __float_as_uint(bv));
        return;
      }
    }
    if (!alpha_pass(*sc, prim, bu, bv)) return;
  }
  atomicMin(&cl.key[owner], mine);
  if (cl.key[owner] == mine) cl.uv[owner] = make_float2(bu, bv);
}
// work the ring off (all 64 lanes of the wave call together); `at_least`: only if it holds that many entries
template <bool ANY_HIT>
FH_D void alpha_flush(const CoopLds& cl, const SceneDev* sc, uint32_t at_least)
{
  uint32_t* const counter = (uint32_t*)(cl.aring + kAlphaRing);
  uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)*counter);
  if (n < at_least || n == 0u) return;
  n = n < kAlphaRing ? n : kAlphaRing;  // (lanes that found the ring full tested in place and left the counter above its size)
  if (__lane_id() < n) {
    const uint4 q = cl.aring[__lane_id()];
    const uint32_t owner = q.x >> 26, prim = q.x & 0x03ffffffu;
    const unsigned long long mine = ((unsigned long long)q.y << 32) | prim;
    const unsigned long long k = cl.key[owner];
    const bool stopped = ANY_HIT && cl.ray[64 + owner].w != 0.0f && (uint32_t)k != 0xffffffffu;
    if (!stopped && mine < k && alpha_pass(*sc, prim, __uint_as_float(q.z), __uint_as_float(q.w))) {
      atomicMin(&cl.key[owner], mine);
      if (cl.key[owner] == mine) cl.uv[owner] = make_float2(__uint_as_float(q.z), __uint_as_float(q.w));
    }
  }
  if (__lane_id() == 0u) *counter = 0u;
}

// MODE 0: every ray wants its closest hit; 1: every ray stops at its first hit; 2: per lane (`any_lane`), as in the streaming kernels
template <int MODE, bool COUNT, bool LDS, bool ALPHA>
FH_D bool traverse_bvh8_coop_mode(const Bvh8Dev& bvh, bool valid, bool any_lane, f3 o, f3 d, float tmax, HitRec& best, uint32_t& n_nodes, uint32_t& n_tris, WaveSteps* ws, const CoopLds& cl,
                                  uint32_t flush, uint2* lds_column, int lds_stride, const SceneDev* sc)
{
  constexpr bool ANY_HIT = MODE != 0;  // (the triangle tests look at the ray's own flag)
  const bool any = MODE == 2 ? any_lane : MODE == 1;
  const uint32_t lane = __lane_id();
  if (!valid) { o = mk3(0.0f); d = mk3(0.0f, 0.0f, 1.0f); tmax = 0.0f; }
  const RayPre rp = ray_prepare(o, d);
  const Ray8 r = ray8_prepare(rp, d);
  cl.ray[lane] = make_float4(rp.o.x, rp.o.y, rp.o.z, rp.Sx);
  cl.ray[64 + lane] = make_float4(rp.Sy, rp.Sz, __uint_as_float((uint32_t)rp.kx | ((uint32_t)rp.ky << 2) | ((uint32_t)rp.kz << 4)), any ? 1.0f : 0.0f);
  const unsigned long long key0 = ((unsigned long long)__float_as_uint(tmax) << 32) | 0xffffffffull;
  cl.key[lane] = key0;
  cl.uv[lane] = make_float2(0.0f, 0.0f);
  GroupStack<LDS> stack(lds_column, lds_stride);
  uint2 group = make_uint2(0u, (valid && bvh.n_nodes) ? 0x80000000u : 0u);
  bool done = !(valid && bvh.n_nodes);
  uint32_t q_head = 0, q_count = 0;  // wave-uniform
  float best_t = tmax;
  for (;;) {
    uint2 tg = make_uint2(0u, 0u);
    if (!done && (group.y & 0xff000000u) == 0u) {
      if (stack.sp == 0) done = true;
      else group = stack.pop();
    }
    if (!done) {
      const unsigned long long k = cl.key[lane];
      best_t = __uint_as_float((uint32_t)(k >> 32));
      if (ANY_HIT && any && (uint32_t)k != 0xffffffffu) done = true;
    }
    if (!done) {
      const uint32_t hits_imask = group.y;
      const uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
      group.y &= ~(1u << bit);
      if (group.y & 0xff000000u) stack.push(group);
      const uint32_t slot = (bit - 24u) ^ r.oct;
      const uint32_t rel = (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
      const uint32_t ni = group.x + rel;
      if (COUNT) { n_nodes++; if (ws && first_active_lane()) ws->node++; }
      node8_visit(bvh, r, ni, best_t, group, tg);
    }
    // hand the candidate triangles to the wave's queue, one per lane and round
    for (;;) {
      const bool has = tg.y != 0u;
      const unsigned long long m = __ballot(has);
      if (m == 0ull) break;
      if (has) {
        const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
        tg.y &= tg.y - 1u;
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        cl.queue[(q_head + q_count + pos) & (kCoopQueue - 1u)] = ((tg.x + b) << 6) | lane;
      }
      q_count += (uint32_t)__popcll(m);
      if (q_count >= 64u) {
        const uint32_t e = cl.queue[(q_head + lane) & (kCoopQueue - 1u)];
        q_head = (q_head + 64u) & (kCoopQueue - 1u);
        q_count -= 64u;
        coop_test<ANY_HIT, COUNT, ALPHA>(bvh, cl, e, n_tris, ws, sc);
      }
    }
    const bool more = __ballot(!done) != 0ull;
    if (q_count && (q_count >= flush || !more)) {
      const uint32_t n = q_count;  // < 64 here
      if (lane < n) {
        const uint32_t e = cl.queue[(q_head + lane) & (kCoopQueue - 1u)];
        coop_test<ANY_HIT, COUNT, ALPHA>(bvh, cl, e, n_tris, ws, sc);
      }
      q_head = (q_head + n) & (kCoopQueue - 1u);
      q_count = 0u;
    }
    if (!more) break;
  }
  const unsigned long long k = cl.key[lane];
  const float2 uv = cl.uv[lane];
  best.t = __uint_as_float((uint32_t)(k >> 32));
  best.u = uv.x; best.v = uv.y;
  best.prim = (uint32_t)k;
  return valid && best.prim != 0xffffffffu;
}
template <bool ANY_HIT, bool COUNT, bool LDS, bool ALPHA>
FH_D bool traverse_bvh8_coop(const Bvh8Dev& bvh, bool valid, f3 o, f3 d, float tmax, HitRec& best, uint32_t& n_nodes, uint32_t& n_tris, WaveSteps* ws, const CoopLds& cl,
                             uint32_t flush, uint2* lds_column, int lds_stride, const SceneDev* sc)
{
  return traverse_bvh8_coop_mode<ANY_HIT ? 1 : 0, COUNT, LDS, ALPHA>(bvh, valid, ANY_HIT, o, d, tmax, best, n_nodes, n_tris, ws, cl, flush, lds_column, lds_stride, sc);
}

// inclusive prefix sum over the 64 lanes of a wave (all lanes active): four row_shr steps inside the rows of 16, then the row totals across the rows (row_bcast:15 / :31).
// Written as v_add_u32 with a DPP operand -- six instructions; from __builtin_amdgcn_update_dpp the compiler makes a v_mov_b32_dpp AND an add per step.  A DPP operand
// written by the VALU instruction before needs two wait states on gfx9 (the assembler does not add them inside an asm block): s_nop 1.
FH_D uint32_t wave_inclusive_sum(uint32_t v)
{
  asm volatile("s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return v;
}

// A wave's share of a work queue: chunks of `chunk` consecutive entries taken from a global cursor.  All members are wave-uniform.
struct ChunkFeed {
  uint32_t* cursor;   // global, zeroed per pass (one word per bounce and kernel)
  uint32_t count, chunk;
  uint32_t cur = 0, end = 0;
  bool exhausted = false;
  FH_D ChunkFeed(uint32_t* c, uint32_t n, uint32_t ch) : cursor(c), count(n), chunk(ch) { exhausted = n == 0u; }
  // position of the first of up to n entries for the asking lanes; entries at or beyond `end` do not exist (yet): those lanes ask again
  FH_D uint32_t reserve(uint32_t n)
  {
    if (cur >= end && !exhausted) {
      uint32_t b = 0;
      if (__lane_id() == 0u) b = atomicAdd(cursor, chunk);
      b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
      cur = b;
      end = b + chunk < count ? b + chunk : count;
      if (b >= count) { exhausted = true; cur = end = 0u; }
    }
    const uint32_t base = cur;
    cur = cur + n < end ? cur + n : end;
    return base;
  }
  FH_D bool drained() const { return exhausted && cur >= end; }
};

// ---------------------------------------------------------------------------------------------
// Streaming form of the wave-cooperative traversal.  Rays differ a lot in length (the longest of 64 is
// ~3x the mean), so a wave that traces one fixed batch runs its node tests with a third of its lanes.
// Here a wave owns a long private sequence of work items and hands a new ray to lanes whose ray has
// finished as soon as `refill` of them are idle.  Work is taken from the launch's queue in chunks (ChunkFeed): one global atomic
// per chunk and wave, nothing per ray, and waves that drew short rays simply draw more chunks, so the launch ends balanced.
// The policy object supplies the work:
//   bool advance(o, d, tmax, any)      lane-local: the next ray of the lane's current item (a path's next secondary ray), if any
//   bool take(i, o, d, tmax, any)      queue entry i (i = pol.feed.reserve(n) + rank among the n asking lanes); false past the chunk's end
//   void commit(hit, h, nodes)         once per ray, after its last candidate was tested
//   bool drained()                     wave-uniform: the queue is used up
//   bool followup()                    the lane's current item may have another ray after the current one
//   ChunkFeed feed                     the wave's share of the queue
// ---------------------------------------------------------------------------------------------
template <bool MIXED, bool COUNT, bool LDS, bool ALPHA, class Policy>
FH_D void traverse_stream(const Bvh8Dev& bvh, Policy& pol, uint32_t& n_nodes, uint32_t& n_tris, WaveSteps* ws, const CoopLds& cl, uint32_t flush, uint32_t refill,
                          uint2* lds_column, int lds_stride, const SceneDev* sc, StackSpill spill = StackSpill{nullptr, 0u}, const uint4* top = nullptr, uint32_t* cost = nullptr)
{
  const uint32_t lane = __lane_id();
  uint32_t cost_node = 0, cost_tri = 0;  // wave-uniform: rounds of node tests / of triangle tests this wave ran (added to cost[0] / cost[1] at the end, one atomic each per wave)
  GroupStack<LDS> stack(lds_column, spill.lds_entries ? (int)spill.lds_entries : lds_stride, spill.area, gridDim.x * blockDim.x, blockIdx.x * blockDim.x + threadIdx.x);
  // ---- Where a ray starts (round 5, profiles/README.md r5-2 / r5-8).  A first-hit ray that leaves a surface may start INSIDE the tree: at the wide node that holds the face
  // it leaves (bvh.parent != null; the policy says which node).  It walks that node's subtree first, then climbs: the parent is visited with the child the ray came up
  // through masked out, its other children are walked with the ordinary stack, and so on to the root -- the nodes a walk from the root visits for a ray that reaches nothing,
  // nearest first, and only the near ones for a ray that is stopped near its origin (1M-triangle soup: mean free path 0.05 in a scene of size 2; a walk from the root pays
  // nine levels for it: 15.3 -> 11.8 node visits per secondary ray).  Hits do not depend on the order nodes are visited in (first-hit: a yes / no), so the bits do not change.
  // State: `up` (the ray still has levels to climb) and entry 0 of the lane's stack, which holds the LINK to climb through next (parent << 3 | child slot, bvh.parent[node]);
  // the groups of the subtree being walked sit above it.  No load depends on another: the link of every node a ray starts at (`fresh`) or climbs to (`climbing`) is fetched
  // NEXT TO that node's own four loads and put into entry 0 after the node test.  (First form, r5-2: the link loaded when the climb happens -- 23 % fewer visits and not a
  // microsecond gained.)  Rays that want their closest hit start at the root: they climb every level anyway.
  // Compiled into the secondary launch of scenes without cut-outs only (kUp): the kernels with the any-hit test have no register for it, and neither has the merged launch of
  // one-pass calls, which carries two policies' state (with the climb and the round counters compiled in it went to 12-32 bytes of scratch and a 16-spp call of configs[3]
  // from 51.6 to 56.2 ms).  Whether a scene starts its rays this way is decided per scene by
  // the host from the round counters below (render.hip; FH_BOTTOM_UP=0 / 1 forces either): dense scenes gain 2-5 % of a frame, interiors of long rays lose 2 %.
  constexpr bool kUp = MIXED && !ALPHA && LDS && Policy::can_climb;
  bool up = false, fresh = false;
  Ray8 r;
  r.o = mk3(0.0f); r.inv = mk3(1.0f); r.oct = 0u; r.nx = r.ny = r.nz = false;
  uint2 group = make_uint2(0u, 0u);
  bool busy = false;  // the lane's ray still has nodes to visit
  bool have = false;  // the lane holds a ray that is not committed yet
  bool any = false;   // the lane's ray stops at the first accepted hit
  uint32_t q_head = 0, q_count = 0;  // wave-uniform
  bool dry = false;                  // wave-uniform: a refill found no work at all
  uint32_t ray_n0 = 0;               // instrumented build: node counter when the lane's ray started
  cl.key[lane] = 0ull;
  for (;;) {
    const unsigned long long idle = __ballot(!busy);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    if (n_idle == 64u || (!dry && n_idle >= refill)) {
      // every candidate of a finished ray must be tested before the ray is committed: drain the queue
      while (q_count) {
        const uint32_t n = q_count < 64u ? q_count : 64u;
        if (kUp) ++cost_tri;
        if (lane < n) coop_test<MIXED, COUNT, ALPHA, AlphaDefer<MIXED, ALPHA>::value>(bvh, cl, cl.queue[(q_head + lane) & (kCoopQueue - 1u)], n_tris, ws, sc);
        q_head = (q_head + n) & (kCoopQueue - 1u);
        q_count -= n;
      }
      if (AlphaDefer<MIXED, ALPHA>::value) alpha_flush<MIXED>(cl, sc, 1u);  // ... and every parked candidate decided
      if (have && !busy) {
        const unsigned long long k = cl.key[lane];
        const float2 uv = cl.uv[lane];
        HitRec h;
        h.t = __uint_as_float((uint32_t)(k >> 32)); h.u = uv.x; h.v = uv.y; h.prim = (uint32_t)k;
        pol.commit(h.prim != 0xffffffffu, h, n_nodes - ray_n0);
        have = false;
      }
      if (!dry) {
        const bool want = !have;
        f3 o = mk3(0.0f), d = mk3(0.0f, 0.0f, 1.0f);
        float tmax = 0.0f;
        bool got = false;
        if (want) got = pol.advance(o, d, tmax, any);  // next ray of the lane's own item, if it has one
        const bool need = want && !got;
        const unsigned long long nm = __ballot(need);
        if (nm != 0ull && !pol.drained()) {
          const uint32_t base = pol.feed.reserve((uint32_t)__popcll(nm));  // wave-uniform
          if (need) got = pol.take(base + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u)), o, d, tmax, any);
        }
        if (got) {
          const RayPre rp = ray_prepare(o, d);
          r = ray8_prepare(rp, d);
          cl.ray[lane] = make_float4(rp.o.x, rp.o.y, rp.o.z, rp.Sx);
          cl.ray[64 + lane] = make_float4(rp.Sy, rp.Sz, __uint_as_float((uint32_t)rp.kx | ((uint32_t)rp.ky << 2) | ((uint32_t)rp.kz << 4)), (MIXED && any) ? 1.0f : 0.0f);
          cl.key[lane] = ((unsigned long long)__float_as_uint(tmax) << 32) | 0xffffffffull;
          cl.uv[lane] = make_float2(0.0f, 0.0f);
          stack.sp = 0;
          have = true;
          const uint32_t start = (kUp && bvh.parent && any) ? pol.start_node() : 0u;  // (a ray that wants its closest hit climbs every level anyway: from the root)
          up = fresh = start != 0u;
          if (up) stack.sp = 1;  // (entry 0: the link to climb through, written when the start node has been visited)
          group = make_uint2(start, 0x80000000u);  // (hit bit 7 of a group without inner-child bits: node group.x itself)
          busy = bvh.n_nodes != 0u;
          if (COUNT) ray_n0 = n_nodes;
        }
        if (__ballot(got) == 0ull && pol.drained() && __ballot(have && pol.followup()) == 0ull) dry = true;
      }
      if (__ballot(busy) == 0ull) {
        if (__ballot(have) == 0ull && dry) break;
        continue;  // only rays without node work (empty BVH) or nothing fetched yet: commit / fetch again
      }
    }
    uint2 tg = make_uint2(0u, 0u);
    float best_t = 0.0f;
    bool walk = busy;
    if (walk) {
      const unsigned long long k = cl.key[lane];
      best_t = __uint_as_float((uint32_t)(k >> 32));
      if (MIXED && any && (uint32_t)k != 0xffffffffu) busy = walk = false;
    }
    bool climbing = false;  // this visit is the parent of the subtree just finished
    if (walk && (group.y & 0xff000000u) == 0u) {
      if (stack.sp == ((kUp && up) ? 1 : 0)) {
        if (!(kUp && up)) busy = walk = false;
        else {
          const uint32_t link = stack.anchor();  // (fetched while the node below was visited)
          if (link == 0xffffffffu) busy = walk = false;  // the root's subtree is done
          else {
            stack.set_anchor(link, link & 7u);
            group = make_uint2(link >> 3, 0x80000000u);
            climbing = true;
          }
        }
      } else group = stack.pop();
    }
    if (kUp && __ballot(walk) != 0ull) ++cost_node;
    if (walk) {
      const uint32_t hits_imask = group.y;
      const uint32_t bit = 31u - (uint32_t)__clz((int)hits_imask);
      group.y &= ~(1u << bit);
      if (group.y & 0xff000000u) stack.push(group);
      constexpr bool ordered = true;  // (slot order for launches whose rays ALL stop at their first hit -- no octant permutation -- was measured and is not faster: tools/patches/r6_pruned_switches.patch)
      const uint32_t slot = ordered ? (bit - 24u) ^ r.oct : bit - 24u;
      const uint32_t ni = group.x + (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
      if (COUNT) { n_nodes++; if (ws && first_active_lane()) ws->node++; }
      uint32_t next_link = 0u;
      if (kUp && (climbing || fresh)) next_link = bvh.parent[ni];  // this node's own way up, in flight next to its four loads
      node8_visit<ordered>(bvh, r, ni, best_t, group, tg, top);
      if (kUp && climbing) group.y &= ~(1u << (24u + (ordered ? (stack.anchor_skip() ^ r.oct) : stack.anchor_skip())));  // the child the ray came up through has been walked
      if (kUp && (climbing || fresh)) { stack.set_anchor_word(next_link); fresh = false; }
    }
    {
    // one wave prefix sum of the lanes' candidate counts (six DPP adds) and a single scatter, instead of one ballot round per candidate of the fullest lane (~15 instructions
    // per round, 2-3 rounds per visit): secondary 842 -> 825 ms per 512 spp of configs[3], 112.6 -> 110.3 ms per configs[2] frame (profiles/README.md r4-1).  The ring holds 128
    // entries, so a visit that would overfill it -- more than 64 candidates on top of a queue below 64 -- falls back to the rounds
    const uint32_t n_cand = (uint32_t)__popc(tg.y);
    const uint32_t incl = wave_inclusive_sum(n_cand);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total != 0u && q_count + total <= kCoopQueue) {
      uint32_t pos = q_head + q_count + incl - n_cand;
      while (tg.y) {
        const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
        tg.y &= tg.y - 1u;
        cl.queue[pos & (kCoopQueue - 1u)] = ((tg.x + b) << 6) | lane;
        ++pos;
      }
      q_count += total;
      while (q_count >= 64u) {
        if (kUp) ++cost_tri;
        coop_test<MIXED, COUNT, ALPHA, AlphaDefer<MIXED, ALPHA>::value>(bvh, cl, cl.queue[(q_head + lane) & (kCoopQueue - 1u)], n_tris, ws, sc);
        q_head = (q_head + 64u) & (kCoopQueue - 1u);
        q_count -= 64u;
      }
    }
    }
    for (;;) {  // (what the scatter above did not take: everything without it, the rare overfull visit with it)
      const bool has = tg.y != 0u;
      const unsigned long long m = __ballot(has);
      if (m == 0ull) break;
      if (has) {
        const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
        tg.y &= tg.y - 1u;
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        cl.queue[(q_head + q_count + pos) & (kCoopQueue - 1u)] = ((tg.x + b) << 6) | lane;
      }
      q_count += (uint32_t)__popcll(m);
      if (q_count >= 64u) {
        if (kUp) ++cost_tri;
        coop_test<MIXED, COUNT, ALPHA, AlphaDefer<MIXED, ALPHA>::value>(bvh, cl, cl.queue[(q_head + lane) & (kCoopQueue - 1u)], n_tris, ws, sc);
        q_head = (q_head + 64u) & (kCoopQueue - 1u);
        q_count -= 64u;
      }
    }
    if (q_count >= flush) {
      const uint32_t n = q_count;  // < 64 here
      if (kUp) ++cost_tri;
      if (lane < n) coop_test<MIXED, COUNT, ALPHA, AlphaDefer<MIXED, ALPHA>::value>(bvh, cl, cl.queue[(q_head + lane) & (kCoopQueue - 1u)], n_tris, ws, sc);
      q_head = (q_head + n) & (kCoopQueue - 1u);
      q_count = 0u;
    }
    if (AlphaDefer<MIXED, ALPHA>::value) alpha_flush<MIXED>(cl, sc, kAlphaFlush);
  }
  if (kUp && cost && lane == 0u) { atomicAdd(cost, cost_node); atomicAdd(cost + 1, cost_tri); }
}

template <bool ANY_HIT, bool COUNT>
FH_D bool traverse(const SceneDev& sc, f3 o, f3 d, float tmax, HitRec& best, uint32_t& n_nodes, uint32_t& n_tris, WaveSteps* ws = nullptr)
{
  if (sc.has_alpha) {  // rare: scenes with cut-out textures take the variant with the any-hit test compiled in
    if (sc.use_bvh8) return traverse_bvh8<ANY_HIT, COUNT, false, true>(sc.bvh8, o, d, tmax, best, n_nodes, n_tris, ws, nullptr, 0, &sc);
    return traverse_bvh2<ANY_HIT, COUNT, true>(sc.bvh2, o, d, tmax, best, n_nodes, n_tris, &sc);
  }
  if (sc.use_bvh8) return traverse_bvh8<ANY_HIT, COUNT, false>(sc.bvh8, o, d, tmax, best, n_nodes, n_tris, ws);
  return traverse_bvh2<ANY_HIT, COUNT>(sc.bvh2, o, d, tmax, best, n_nodes, n_tris);
}

}  // namespace fh
