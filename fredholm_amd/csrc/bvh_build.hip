// bvh_build.hip -- on-device BVH construction.
//
// Replaces Renderer::build_gas / build_ias (fredholm/include/fredholm/renderer.h:434-552), i.e. the
// closed-source optixAccelBuild.  Pipeline, all on the context stream:
//   1. per-face world-space bounds + scene bounds              (k_face_bounds)
//   2. 63-bit Morton code of the box centre                    (k_morton)
//   3. rocPRIM radix sort of (code, face)                      (sort)
//   4. Karras 2012 radix-tree hierarchy, one thread per node   (k_hierarchy)
//   5. bottom-up box refit with arrival counters               (k_refit)
//   6. emit the traversal layout: 64-byte two-child nodes whose children are inner nodes or leaves of
//      <= 4 Morton-contiguous triangles, and the triangle array in leaf order (k_emit2 / k_emit_tris)
//   6a. optionally PLOC instead of the radix tree (chosen per scene by summed inner-node area) and (round 5, measured and taken out: tools/patches) SAH refinement by parallel reinsertion (FH_SAH_ITERS; off: it lowers
//      the area and not the visits, profiles/README.md r5-1)
//   7. collapse to the 8-wide quantised layout the traversal kernels prefer (bvh8 section), with the way up -- per wide node parent << 3 | child slot, per face the node
//      that holds it (also written into the face record) -- for rays that start at their face (fh_trace.h)
// Triangle order is the Morton order, so leaf reads are contiguous 48-byte records.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cstdlib>

#include "context.h"
#include "fh_trace.h"

namespace fh {

namespace {

constexpr uint32_t kLeafMax2 = 4;

__device__ __forceinline__ int float_order(float f)
{
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float order_float(int i)
{
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &j, 4);
  return f;
}

// bounds[0..2] = min (ordered ints), bounds[3..5] = max
__global__ void k_face_bounds(const float4* face_rec, uint32_t n, float4* lo, float4* hi, int* bounds)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  float l[3] = {3e38f, 3e38f, 3e38f}, h[3] = {-3e38f, -3e38f, -3e38f};
  if (f < n) {
    const float4 a = face_rec[kFaceRec * (size_t)f], b = face_rec[kFaceRec * (size_t)f + 1], c = face_rec[kFaceRec * (size_t)f + 2];
    l[0] = fminf(a.x, fminf(b.x, c.x)); l[1] = fminf(a.y, fminf(b.y, c.y)); l[2] = fminf(a.z, fminf(b.z, c.z));
    h[0] = fmaxf(a.x, fmaxf(b.x, c.x)); h[1] = fmaxf(a.y, fmaxf(b.y, c.y)); h[2] = fmaxf(a.z, fmaxf(b.z, c.z));
    lo[f] = make_float4(l[0], l[1], l[2], 0.0f);
    hi[f] = make_float4(h[0], h[1], h[2], 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float mn = l[k], mx = h[k];
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&bounds[k], float_order(mn));
      atomicMax(&bounds[3 + k], float_order(mx));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Early split clipping (Ernst & Greiner, "Early Split Clipping for Bounding Volume Hierarchies", RT 2007).  A triangle that is
// large against the scene (a floor, a wall, a cable) has a bounding box that most rays enter without coming near the triangle.
// Such a triangle enters the build as several REFERENCES instead: its box is cut into a grid along its two longest axes, the
// triangle is clipped to every cell and each non-empty cell becomes a leaf with the tight box of the clipped polygon.  All
// references of a face store the same triangle; a ray that meets two of them computes the identical (t, face) pair, which the
// closest-hit order treats as one hit.  Faces below the size threshold keep their single box, so a scene without large triangles
// (the bench soup) builds exactly as before.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxSplitCells = 64;  // references per face at most

struct SplitGrid { int a, b, ka, kb; };  // axes (a = longest) and cells per axis

__device__ __forceinline__ float axis_of(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

__device__ inline SplitGrid split_grid(const float4& lo, const float4& hi, float thr)
{
  const float e[3] = {hi.x - lo.x, hi.y - lo.y, hi.z - lo.z};
  int a = 0;
  if (e[1] > e[a]) a = 1;
  if (e[2] > e[a]) a = 2;
  int b = (a + 1) % 3;
  const int c = (a + 2) % 3;
  if (e[c] > e[b]) b = c;
  SplitGrid g{a, b, 1, 1};
  if (!(e[a] > thr)) return g;
  float fa = ceilf(e[a] / thr), fb = e[b] > thr ? ceilf(e[b] / thr) : 1.0f;
  while (fa * fb > (float)kMaxSplitCells) { if (fa >= fb) fa = ceilf(fa * 0.5f); else fb = ceilf(fb * 0.5f); }
  g.ka = (int)fa; g.kb = (int)fb;
  return g;
}

// clip the polygon (<= 8 vertices) against the half-space  sign * (p[axis] - plane) >= 0
__device__ inline int clip_poly(float (*p)[3], int n, int axis, float plane, float sign)
{
  float out[9][3];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* cur = p[i];
    const float* nxt = p[(i + 1) % n];
    const float dc = sign * (cur[axis] - plane), dn = sign * (nxt[axis] - plane);
    if (dc >= 0.0f) { out[m][0] = cur[0]; out[m][1] = cur[1]; out[m][2] = cur[2]; ++m; }
    if ((dc >= 0.0f) != (dn >= 0.0f)) {
      const float t = dc / (dc - dn);
      out[m][0] = cur[0] + t * (nxt[0] - cur[0]); out[m][1] = cur[1] + t * (nxt[1] - cur[1]); out[m][2] = cur[2] + t * (nxt[2] - cur[2]);
      out[m][axis] = plane;
      ++m;
    }
  }
  for (int i = 0; i < m; ++i) { p[i][0] = out[i][0]; p[i][1] = out[i][1]; p[i][2] = out[i][2]; }
  return m;
}

// box of the part of triangle (v0, v1, v2) inside cell (i, j) of the grid; false if the cell is empty
__device__ inline bool split_cell_box(const float4& v0, const float4& v1, const float4& v2, const float4& lo, const float4& hi, const SplitGrid& g, int i, int j, float eps,
                                      float4& clo, float4& chi)
{
  float poly[9][3] = {{v0.x, v0.y, v0.z}, {v1.x, v1.y, v1.z}, {v2.x, v2.y, v2.z}};
  int n = 3;
  const float la = axis_of(lo, g.a), ha = axis_of(hi, g.a), lb = axis_of(lo, g.b), hb = axis_of(hi, g.b);
  const float a0 = la + (ha - la) * ((float)i / (float)g.ka), a1 = i + 1 == g.ka ? ha : la + (ha - la) * ((float)(i + 1) / (float)g.ka);
  const float b0 = lb + (hb - lb) * ((float)j / (float)g.kb), b1 = j + 1 == g.kb ? hb : lb + (hb - lb) * ((float)(j + 1) / (float)g.kb);
  n = clip_poly(poly, n, g.a, a0, 1.0f); if (n < 3) return false;
  n = clip_poly(poly, n, g.a, a1, -1.0f); if (n < 3) return false;
  if (g.kb > 1) {
    n = clip_poly(poly, n, g.b, b0, 1.0f); if (n < 3) return false;
    n = clip_poly(poly, n, g.b, b1, -1.0f); if (n < 3) return false;
  }
  float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
  for (int k = 0; k < n; ++k)
    for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], poly[k][c]); mx[c] = fmaxf(mx[c], poly[k][c]); }
  // a little slack for the rounding of the clip points, but never beyond the triangle's own box
  clo = make_float4(fmaxf(mn[0] - eps, lo.x), fmaxf(mn[1] - eps, lo.y), fmaxf(mn[2] - eps, lo.z), 0.0f);
  chi = make_float4(fminf(mx[0] + eps, hi.x), fminf(mx[1] + eps, hi.y), fminf(mx[2] + eps, hi.z), 0.0f);
  return true;
}

// pass 1: references per face (1 = unsplit)
__global__ void k_split_count(const float4* face_rec, const float4* face_lo, const float4* face_hi, uint32_t n, float thr, float eps, uint32_t* count)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float4 lo = face_lo[f], hi = face_hi[f];
  const SplitGrid g = split_grid(lo, hi, thr);
  uint32_t c = 1;
  if (g.ka * g.kb > 1) {
    const float4 v0 = face_rec[kFaceRec * (size_t)f], v1 = face_rec[kFaceRec * (size_t)f + 1], v2 = face_rec[kFaceRec * (size_t)f + 2];
    c = 0;
    float4 a, b;
    for (int i = 0; i < g.ka; ++i)
      for (int j = 0; j < g.kb; ++j)
        if (split_cell_box(v0, v1, v2, lo, hi, g, i, j, eps, a, b)) ++c;
    if (c == 0) c = 1;  // degenerate triangle: keep its box
  }
  count[f] = c;
}

// pass 2: write the references
__global__ void k_split_emit(const float4* face_rec, const float4* face_lo, const float4* face_hi, uint32_t n, float thr, float eps, const uint32_t* count, const uint32_t* offset,
                             uint32_t* ref_face, float4* ref_lo, float4* ref_hi)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float4 lo = face_lo[f], hi = face_hi[f];
  uint32_t o = offset[f];
  const SplitGrid g = split_grid(lo, hi, thr);
  if (g.ka * g.kb > 1 && count[f] > 0) {
    const float4 v0 = face_rec[kFaceRec * (size_t)f], v1 = face_rec[kFaceRec * (size_t)f + 1], v2 = face_rec[kFaceRec * (size_t)f + 2];
    uint32_t written = 0;
    float4 a, b;
    for (int i = 0; i < g.ka; ++i)
      for (int j = 0; j < g.kb; ++j)
        if (split_cell_box(v0, v1, v2, lo, hi, g, i, j, eps, a, b)) { ref_face[o] = f; ref_lo[o] = a; ref_hi[o] = b; ++o; ++written; }
    if (written) return;
  }
  ref_face[o] = f; ref_lo[o] = lo; ref_hi[o] = hi;
}

__device__ __forceinline__ unsigned long long expand21(unsigned long long v)
{
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

__global__ void k_morton(const float4* lo, const float4* hi, uint32_t n, const int* bounds, unsigned long long* keys, uint32_t* vals)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float bl[3] = {order_float(bounds[0]), order_float(bounds[1]), order_float(bounds[2])};
  const float bh[3] = {order_float(bounds[3]), order_float(bounds[4]), order_float(bounds[5])};
  const float4 l = lo[f], h = hi[f];
  const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
  unsigned long long q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ext = bh[k] - bl[k];
    float x = ext > 0.0f ? (c[k] - bl[k]) / ext : 0.0f;
    x = fminf(fmaxf(x * 2097152.0f, 0.0f), 2097151.0f);
    q[k] = (unsigned long long)x;
  }
  keys[f] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
  vals[f] = f;
}

__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j)
{
  if (j < 0 || j >= n) return -1;
  const unsigned long long x = keys[i] ^ keys[j];
  if (x == 0ull) return 64 + __clz((unsigned)(i ^ j));
  return __clzll((long long)x);
}

// child reference: >= 0 inner node, < 0 leaf (~ref = sorted position)
__global__ void k_hierarchy(const unsigned long long* keys, int n, int2* children, int2* ranges, int* node_parent, int* leaf_parent)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const int left = (first == gamma) ? ~gamma : gamma;
  const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  children[i] = make_int2(left, right);
  ranges[i] = make_int2(first, last);
  if (left >= 0) node_parent[left] = i; else leaf_parent[~left] = i;
  if (right >= 0) node_parent[right] = i; else leaf_parent[~right] = i;
  if (i == 0) node_parent[0] = -1;
}

__device__ __forceinline__ void child_box(int ref, const float4* node_lo, const float4* node_hi, const float4* leaf_lo, const float4* leaf_hi, float4& lo, float4& hi)
{
  if (ref >= 0) { lo = node_lo[ref]; hi = node_hi[ref]; }
  else { lo = leaf_lo[~ref]; hi = leaf_hi[~ref]; }
}

// bottom-up refit: the second thread to arrive at a node merges the two child boxes
__global__ void k_refit(const uint32_t* sorted_face, const float4* face_lo, const float4* face_hi, int n, const int2* children, const int* node_parent, const int* leaf_parent,
                        float4* node_lo, float4* node_hi, float4* leaf_lo, float4* leaf_hi, unsigned int* arrive)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = sorted_face[i];
  leaf_lo[i] = face_lo[f];
  leaf_hi[i] = face_hi[f];
  int cur = leaf_parent[i];
  while (cur >= 0) {
    __threadfence();  // publish this subtree's box before signalling
    const unsigned int prev = atomicAdd(&arrive[cur], 1u);
    if (prev == 0u) return;
    __threadfence();  // see the sibling's box
    const int2 ch = children[cur];
    float4 al, ah, bl, bh;
    child_box(ch.x, node_lo, node_hi, leaf_lo, leaf_hi, al, ah);
    child_box(ch.y, node_lo, node_hi, leaf_lo, leaf_hi, bl, bh);
    node_lo[cur] = make_float4(fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), 0.0f);
    node_hi[cur] = make_float4(fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), 0.0f);
    cur = node_parent[cur];
  }
}

// traversal reference of a radix-tree child: small subtrees become leaves
__device__ __forceinline__ int emit_ref(int ref, const int2* ranges)
{
  if (ref < 0) return ~(int)(((uint32_t)(~ref) << 3) | 0u);
  const int2 r = ranges[ref];
  const uint32_t cnt = (uint32_t)(r.y - r.x + 1);
  if (cnt <= kLeafMax2) return ~(int)(((uint32_t)r.x << 3) | (cnt - 1u));
  return ref;
}

__global__ void k_emit2(int n_inner, const int2* children, const int2* ranges, const float4* node_lo, const float4* node_hi, const float4* leaf_lo, const float4* leaf_hi, float pad,
                        float4* out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  const int2 ch = children[i];
  float4 al, ah, bl, bh;
  child_box(ch.x, node_lo, node_hi, leaf_lo, leaf_hi, al, ah);
  child_box(ch.y, node_lo, node_hi, leaf_lo, leaf_hi, bl, bh);
  const int ra = emit_ref(ch.x, ranges), rb = emit_ref(ch.y, ranges);
  out[4 * i] = make_float4(al.x - pad, al.y - pad, al.z - pad, ah.x + pad);
  out[4 * i + 1] = make_float4(ah.y + pad, ah.z + pad, bl.x - pad, bl.y - pad);
  out[4 * i + 2] = make_float4(bl.z - pad, bh.x + pad, bh.y + pad, bh.z + pad);
  out[4 * i + 3] = make_float4(__int_as_float(ra), __int_as_float(rb), 0.0f, 0.0f);
}

// root for scenes with <= kLeafMax2 faces: child 0 is the only leaf, child 1 is an empty box
__global__ void k_emit2_tiny(int n, const float4* face_lo, const float4* face_hi, float pad, float4* out)
{
  float l[3] = {3e38f, 3e38f, 3e38f}, h[3] = {-3e38f, -3e38f, -3e38f};
  for (int f = 0; f < n; ++f) {
    l[0] = fminf(l[0], face_lo[f].x); l[1] = fminf(l[1], face_lo[f].y); l[2] = fminf(l[2], face_lo[f].z);
    h[0] = fmaxf(h[0], face_hi[f].x); h[1] = fmaxf(h[1], face_hi[f].y); h[2] = fmaxf(h[2], face_hi[f].z);
  }
  out[0] = make_float4(l[0] - pad, l[1] - pad, l[2] - pad, h[0] + pad);
  out[1] = make_float4(h[1] + pad, h[2] + pad, 3e38f, 3e38f);
  out[2] = make_float4(3e38f, 3e38f, 3e38f, 3e38f);  // unreachable point box: every slab test fails
  const int ra = ~(int)((0u << 3) | (uint32_t)(n - 1));
  out[3] = make_float4(__int_as_float(ra), __int_as_float(ra), 0.0f, 0.0f);
}

__global__ void k_emit_tris(const float4* face_rec, const uint8_t* face_cls, const uint32_t* sorted_face, uint32_t n, float4* tris, const uint32_t* ref_face)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t f = sorted_face[i];
  if (ref_face) f = ref_face[f];  // split faces: the leaf is a reference
  if (face_cls[f] & 0x20u) {  // a cut-out face whose any-hit test can never pass (capi.hip: footprint_class): a triangle no ray hits
    tris[3 * i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
    tris[3 * i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    tris[3 * i + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  const float4 a = face_rec[kFaceRec * (size_t)f], b = face_rec[kFaceRec * (size_t)f + 1], c = face_rec[kFaceRec * (size_t)f + 2];
  tris[3 * i] = make_float4(a.x, a.y, a.z, __uint_as_float(f));
  tris[3 * i + 1] = make_float4(b.x, b.y, b.z, (face_cls[f] & 0x40u) ? 1.0f : 0.0f);  // .w != 0: candidate hits need the alpha test
  tris[3 * i + 2] = make_float4(c.x, c.y, c.z, 0.0f);
}


// ------------------------------------------------------------------------------------------------
// BVH8 collapse.  One thread per wide node: starting from the two children of a radix-tree node it
// repeatedly opens the child with the largest surface area (among those holding more than
// kLeafMax8 triangles) until eight children exist, assigns children to slots so that slot bits
// agree with the octant of (child centre - node centre), quantises the boxes and hands the inner
// children to the next level.  Levels are processed breadth first; node and triangle blocks are
// allocated with atomics (the traversal result does not depend on their order, fh_trace.h).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kLeafMax8 = 3;  // scenes with at most this many faces get a single root node (k_collapse8_tiny)

struct Work8 { int bnode; uint32_t wnode; };

__device__ __forceinline__ float box_area(const float4& lo, const float4& hi)
{
  const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
  return ex * ey + ey * ez + ez * ex;
}

__device__ __forceinline__ uint32_t ref_count(int ref, const int2* ranges) { return ref < 0 ? 1u : (uint32_t)(ranges[ref].y - ranges[ref].x + 1); }
__device__ __forceinline__ uint32_t ref_first(int ref, const int2* ranges) { return ref < 0 ? (uint32_t)(~ref) : (uint32_t)ranges[ref].x; }

// smallest biased exponent E with 255 * 2^(E-127) >= extent
__device__ __forceinline__ uint32_t quant_exponent(float extent)
{
  if (!(extent > 0.0f)) return 1u;
  const float s = extent / 255.0f;
  uint32_t bits = __float_as_uint(s);
  uint32_t e = (bits >> 23) & 0xffu;
  if (bits & 0x7fffffu) e += 1u;  // round the scale up to a power of two
  if (e < 1u) e = 1u;
  if (e > 254u) e = 254u;
  // guard against rounding in the division above
  while (e < 254u && __uint_as_float(e << 23) * 255.0f < extent) e += 1u;
  return e;
}

// Origin and scale of one axis of a wide node share a word (fh_trace.h): the low mantissa byte of the origin holds the biased exponent of the
// scale, and the origin IS the float the word spells.  Returns the word for lower bound `lo` and upper bound `hi`: the largest such float that
// is <= lo, with the smallest exponent whose 255 steps still reach hi from there.
__device__ __forceinline__ uint32_t origin_word(float lo, uint32_t e)
{
  const uint32_t b = __float_as_uint(lo);
  if (!(b >> 31)) {                       // lo >= +0: clearing mantissa bits rounds down, the exponent byte may round up again
    uint32_t w = (b & ~0xffu) | e;
    if (w > b) w = w >= 0x100u ? w - 0x100u : (0x80000000u | e);  // (nothing below on the positive side: the negative number closest to zero)
    return w;
  }
  const uint32_t m = b & 0x7fffffffu;     // lo <= -0: the magnitude has to reach |lo|
  uint32_t wm = (m & ~0xffu) | e;
  if (wm < m) wm += 0x100u;
  return 0x80000000u | wm;
}
__device__ __forceinline__ uint32_t quant_axis(float lo, float hi, float& origin)
{
  uint32_t e = quant_exponent(hi - lo);
  for (;;) {
    const uint32_t w = origin_word(lo, e);
    origin = __uint_as_float(w);
    if (e >= 254u || __uint_as_float(e << 23) * 255.0f >= hi - origin) return w;
    e += 1u;
  }
}

// ---- which descendants of a binary node become the children of its wide node: the cut of least cost (Ylitie, Karras, Laine 2017, section 3).
// With one triangle per leaf child the triangles' share of the SAH cost is the same for every cut, so what is minimised is the summed surface area of the
// wide nodes, i.e. the expected number of 8-wide node tests per ray.  C(n, i) = least cost of the subtree of n when it may occupy at most i child slots
// of the wide node above it:
//   C(n, 1) = area(n) + D(n, 8)                n becomes a wide node itself and deals its eight slots to its two children
//   D(n, j) = min over k of C(left, k) + C(right, j - k)
//   C(n, i) = min(D(n, i), C(n, i - 1))        i = 2..7; a triangle costs nothing and takes one slot
// Tables are filled bottom up (arrival counters, like the box refit); per node 7 costs and the decisions: the k of D(n, j) for j = 2..8 (3 bits each in x)
// and the number of slots C(n, i) really uses for i = 1..7 (3 bits each in y).
__global__ void k_tree_parents(int n_inner, const int2* children, int* node_parent, int* leaf_parent)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  const int2 ch = children[i];
  if (ch.x >= 0) node_parent[ch.x] = i; else leaf_parent[~ch.x] = i;
  if (ch.y >= 0) node_parent[ch.y] = i; else leaf_parent[~ch.y] = i;
}

__device__ __forceinline__ float cut_cost(int ref, int k, const float* cost) { return ref < 0 ? 0.0f : cost[7 * (size_t)ref + (k - 1)]; }

__global__ void k_cut_tables(int n_leaves, const int2* children, const int* node_parent, const int* leaf_parent, const float4* node_lo, const float4* node_hi, unsigned int* arrive,
                             float* cost, uint2* decision)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_leaves) return;
  int cur = leaf_parent[i];
  while (cur >= 0) {
    __threadfence();  // publish this subtree's table before signalling
    const unsigned int prev = atomicAdd(&arrive[cur], 1u);
    if (prev == 0u) return;
    __threadfence();  // see the sibling's table
    const int2 ch = children[cur];
    float D[9];
    uint32_t split = 0, use = 1u;
    for (int j = 2; j <= 8; ++j) {
      float best = 3e38f;
      int bk = 1;
      for (int k = (j - 7 > 1 ? j - 7 : 1); k <= (j - 1 < 7 ? j - 1 : 7); ++k) {
        const float c = cut_cost(ch.x, k, cost) + cut_cost(ch.y, j - k, cost);
        if (c < best) { best = c; bk = k; }
      }
      D[j] = best;
      split |= (uint32_t)bk << (3 * (j - 2));
    }
    float C = box_area(node_lo[cur], node_hi[cur]) + D[8];
    uint32_t u = 1u;
    cost[7 * (size_t)cur] = C;
    for (int k = 2; k <= 7; ++k) {
      if (D[k] < C) { C = D[k]; u = (uint32_t)k; }
      cost[7 * (size_t)cur + (k - 1)] = C;
      use |= u << (3 * (k - 1));
    }
    decision[cur] = make_uint2(split, use);
    cur = node_parent[cur];
  }
}

__global__ void k_collapse8(const Work8* items, uint32_t n_items, const int2* children, const int2* ranges, const float4* node_lo, const float4* node_hi, const float4* leaf_lo,
                            const float4* leaf_hi, float pad, uint32_t leaf_max, uint32_t absorb, uint4* nodes, uint32_t* node_counter, uint32_t* tri_counter, uint32_t* tri_slot, Work8* next_items, uint32_t* next_count,
                            const uint2* decision, uint32_t* wparent)
{
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_items) return;
  const Work8 it = items[w];
  int ref[8];
  float4 lo[8], hi[8];
  int n = 0;
  {
    const int2 ch = children[it.bnode];
    ref[0] = ch.x; ref[1] = ch.y; n = 2;
    child_box(ch.x, node_lo, node_hi, leaf_lo, leaf_hi, lo[0], hi[0]);
    child_box(ch.y, node_lo, node_hi, leaf_lo, leaf_hi, lo[1], hi[1]);
  }
  // Which child to open next.  Plain rule: the largest one.  With `absorb` (default) small subtrees are treated as units: a child
  // with at most 8 leaves is opened only when ALL of its leaves fit into the free slots, and is then opened completely before
  // anything else, so it ends up either as direct leaf children of this node or as ONE full inner child below it, never as a few
  // 2-3 leaf fragments that each cost a node fetch (the plain rule left 4.5 of 8 slots used on average).
  if (decision) {  // the cut chosen by k_cut_tables: deal the eight slots down the binary tree
    int st_ref[8], st_slots[8], sp = 0;
    const int2 ch0 = children[it.bnode];
    const int k0 = (int)((decision[it.bnode].x >> 18) & 7u);
    st_ref[sp] = ch0.y; st_slots[sp++] = 8 - k0;
    st_ref[sp] = ch0.x; st_slots[sp++] = k0;
    n = 0;
    while (sp > 0) {
      --sp;
      const int m = st_ref[sp], j = st_slots[sp];
      if (m < 0) { ref[n++] = m; continue; }
      const uint2 d = decision[m];
      const int jj = (int)((d.y >> (3 * (j - 1))) & 7u);  // slots the subtree really uses of the j it was given
      if (jj <= 1) { ref[n++] = m; continue; }            // one slot: it becomes a wide node of its own
      const int k = (int)((d.x >> (3 * (jj - 2))) & 7u);
      const int2 ch = children[m];
      st_ref[sp] = ch.y; st_slots[sp++] = jj - k;
      st_ref[sp] = ch.x; st_slots[sp++] = k;
    }
    for (int i = 0; i < n; ++i) child_box(ref[i], node_lo, node_hi, leaf_lo, leaf_hi, lo[i], hi[i]);
  }
  uint32_t absorbing = 0;  // slots that belong to a subtree being absorbed
  while (!decision && n < 8) {
    int best = -1;
    float best_area = -1.0f;
    if (absorb) {
      for (int i = 0; i < n && best < 0; ++i)
        if ((absorbing >> i) & 1u) { if (ref[i] >= 0 && ref_count(ref[i], ranges) > leaf_max) best = i; else absorbing &= ~(1u << i); }
      if (best < 0) {
        for (int i = 0; i < n; ++i) {  // subtrees that cannot become one node: open the largest
          if (ref[i] < 0 || ref_count(ref[i], ranges) <= 8u * leaf_max) continue;
          const float a = box_area(lo[i], hi[i]);
          if (a > best_area) { best_area = a; best = i; }
        }
      }
      if (best < 0) {
        for (int i = 0; i < n; ++i) {  // small subtrees whose leaves all fit: absorb the largest
          if (ref[i] < 0) continue;
          const uint32_t c = ref_count(ref[i], ranges);
          if (c <= leaf_max || (uint32_t)n + (c + leaf_max - 1u) / leaf_max - 1u > 8u) continue;
          const float a = box_area(lo[i], hi[i]);
          if (a > best_area) { best_area = a; best = i; }
        }
        if (best >= 0) absorbing |= 1u << best;
      }
    } else {
      for (int i = 0; i < n; ++i) {
        if (ref[i] < 0 || ref_count(ref[i], ranges) <= leaf_max) continue;
        const float a = box_area(lo[i], hi[i]);
        if (a > best_area) { best_area = a; best = i; }
      }
    }
    if (best < 0) break;
    if ((absorbing >> best) & 1u) absorbing |= 1u << n;
    const int2 ch = children[ref[best]];
    ref[best] = ch.x;
    child_box(ch.x, node_lo, node_hi, leaf_lo, leaf_hi, lo[best], hi[best]);
    ref[n] = ch.y;
    child_box(ch.y, node_lo, node_hi, leaf_lo, leaf_hi, lo[n], hi[n]);
    ++n;
  }
  // node box = union of the padded child boxes
  float nlo[3] = {3e38f, 3e38f, 3e38f}, nhi[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = 0; i < n; ++i) {
    lo[i].x -= pad; lo[i].y -= pad; lo[i].z -= pad; hi[i].x += pad; hi[i].y += pad; hi[i].z += pad;
    nlo[0] = fminf(nlo[0], lo[i].x); nlo[1] = fminf(nlo[1], lo[i].y); nlo[2] = fminf(nlo[2], lo[i].z);
    nhi[0] = fmaxf(nhi[0], hi[i].x); nhi[1] = fmaxf(nhi[1], hi[i].y); nhi[2] = fmaxf(nhi[2], hi[i].z);
  }
  const float cx = 0.5f * (nlo[0] + nhi[0]), cy = 0.5f * (nlo[1] + nhi[1]), cz = 0.5f * (nlo[2] + nhi[2]);
  // greedy octant assignment: repeatedly take the (child, slot) pair with the largest projection
  int slot_of[8], child_in[8];
  for (int i = 0; i < 8; ++i) { slot_of[i] = -1; child_in[i] = -1; }
  for (int round = 0; round < n; ++round) {
    float bestv = -3e38f;
    int bc = -1, bs = -1;
    for (int i = 0; i < n; ++i) {
      if (slot_of[i] >= 0) continue;
      const float dx = 0.5f * (lo[i].x + hi[i].x) - cx, dy = 0.5f * (lo[i].y + hi[i].y) - cy, dz = 0.5f * (lo[i].z + hi[i].z) - cz;
      for (int sl = 0; sl < 8; ++sl) {
        if (child_in[sl] >= 0) continue;
        const float v = ((sl & 4) ? dx : -dx) + ((sl & 2) ? dy : -dy) + ((sl & 1) ? dz : -dz);
        if (v > bestv) { bestv = v; bc = i; bs = sl; }
      }
    }
    slot_of[bc] = bs;
    child_in[bs] = bc;
  }
  // count inner children / triangles, allocate the block of inner children (triangles live in the node's own eight slots)
  uint32_t n_inner = 0, n_tris = 0;
  for (int sl = 0; sl < 8; ++sl) {
    const int c = child_in[sl];
    if (c < 0) continue;
    const uint32_t cnt = ref_count(ref[c], ranges);
    if (ref[c] >= 0 && cnt > leaf_max) n_inner++; else n_tris += cnt;
  }
  const uint32_t child_base = n_inner ? atomicAdd(node_counter, n_inner) : 0u;
  if (n_tris) atomicAdd(tri_counter, n_tris);
  float org[3];
  const uint32_t wx = quant_axis(nlo[0], nhi[0], org[0]), wy = quant_axis(nlo[1], nhi[1], org[1]), wz = quant_axis(nlo[2], nhi[2], org[2]);
  const float isx = 1.0f / __uint_as_float((wx & 0xffu) << 23), isy = 1.0f / __uint_as_float((wy & 0xffu) << 23), isz = 1.0f / __uint_as_float((wz & 0xffu) << 23);
  uint32_t imask = 0, q[6][2] = {{~0u, ~0u}, {~0u, ~0u}, {~0u, ~0u}, {0, 0}, {0, 0}, {0, 0}};  // empty slots keep the inverted box (lo 255, hi 0)
  uint32_t inner_seen = 0;
  uint32_t next_base = 0;
  if (n_inner) next_base = atomicAdd(next_count, n_inner);
  for (int sl = 0; sl < 8; ++sl) {
    const int c = child_in[sl];
    if (c < 0) continue;
    const uint32_t cnt = ref_count(ref[c], ranges);
    if (ref[c] >= 0 && cnt > leaf_max) {
      imask |= 1u << sl;
      next_items[next_base + inner_seen] = Work8{ref[c], child_base + inner_seen};
      wparent[child_base + inner_seen] = (it.wnode << 3) | (uint32_t)sl;  // the way up (fh_trace.h: bottom-up start of rays that leave a surface)
      inner_seen++;
    } else {
      tri_slot[ref_first(ref[c], ranges)] = 8u * it.wnode + (uint32_t)sl;  // (leaf_max is 1: one triangle per leaf child)
    }
    const float v[6] = {floorf((lo[c].x - org[0]) * isx), floorf((lo[c].y - org[1]) * isy), floorf((lo[c].z - org[2]) * isz),
                        ceilf((hi[c].x - org[0]) * isx), ceilf((hi[c].y - org[1]) * isy), ceilf((hi[c].z - org[2]) * isz)};
    for (int k = 0; k < 6; ++k) {
      const float cl = fminf(fmaxf(v[k], 0.0f), 255.0f);
      q[k][sl >> 2] = (q[k][sl >> 2] & ~(0xffu << (8 * (sl & 3)))) | (((uint32_t)cl) << (8 * (sl & 3)));
    }
  }
  uint4* out = nodes + kBvh8NodeVec * (size_t)it.wnode;
  out[0] = make_uint4(wx, wy, wz, (child_base << 8) | imask);
  out[1] = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
  out[2] = make_uint4(q[2][0], q[2][1], q[3][0], q[3][1]);
  out[3] = make_uint4(q[4][0], q[4][1], q[5][0], q[5][1]);
}

// scenes with <= kLeafMax8 faces: a root with one leaf child per face
__global__ void k_collapse8_tiny(int n, const float4* face_lo, const float4* face_hi, float pad, uint4* nodes, uint32_t* tri_slot)
{
  float l[3] = {3e38f, 3e38f, 3e38f}, h[3] = {-3e38f, -3e38f, -3e38f};
  for (int f = 0; f < n; ++f) {
    l[0] = fminf(l[0], face_lo[f].x - pad); l[1] = fminf(l[1], face_lo[f].y - pad); l[2] = fminf(l[2], face_lo[f].z - pad);
    h[0] = fmaxf(h[0], face_hi[f].x + pad); h[1] = fmaxf(h[1], face_hi[f].y + pad); h[2] = fmaxf(h[2], face_hi[f].z + pad);
    tri_slot[f] = (uint32_t)f;
  }
  float org[3];
  const uint32_t wx = quant_axis(l[0], h[0], org[0]), wy = quant_axis(l[1], h[1], org[1]), wz = quant_axis(l[2], h[2], org[2]);
  // every face gets the whole node box (0 .. 255 on every axis): with <= 3 triangles nothing is gained by tighter ones
  const uint32_t lo4 = n >= 4 ? 0u : (~0u << (8 * n)), hi4 = ~lo4;
  nodes[0] = make_uint4(wx, wy, wz, 0u);
  nodes[1] = make_uint4(lo4, ~0u, lo4, ~0u);
  nodes[2] = make_uint4(lo4, ~0u, hi4, 0u);
  nodes[3] = make_uint4(hi4, 0u, hi4, 0u);
}

// triangle slots of the wide tree: 8 per node, slot 8 * node + child slot; slots without a triangle hold a degenerate one (all zero, face id 0xffffffff)
__global__ void k_emit_tris8(const float4* face_rec, const uint8_t* face_cls, const uint32_t* sorted_face, const uint32_t* tri_slot, uint32_t n, float4* tris, const uint32_t* ref_face,
                             uint32_t* face_node, const uint32_t* split_count)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t f = sorted_face[i];
  if (ref_face) f = ref_face[f];
  if (face_node) face_node[f] = (split_count && split_count[f] > 1u) ? 0u : tri_slot[i] >> 3;  // the wide node that holds the face; a face that entered the build as several references: the root
  const size_t t = 3 * (size_t)tri_slot[i];
  if (face_cls[f] & 0x20u) return;  // never hit (above): the slot keeps the degenerate triangle k_clear_tris8 put there
  const float4 a = face_rec[kFaceRec * (size_t)f], b = face_rec[kFaceRec * (size_t)f + 1], c = face_rec[kFaceRec * (size_t)f + 2];
  tris[t] = make_float4(a.x, a.y, a.z, __uint_as_float(f));
  tris[t + 1] = make_float4(b.x, b.y, b.z, (face_cls[f] & 0x40u) ? 1.0f : 0.0f);
  tris[t + 2] = make_float4(c.x, c.y, c.z, 0.0f);
}
__global__ void k_clear_tris8(float4* tris, uint32_t n_slots)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  tris[3 * (size_t)i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xffffffffu));
  tris[3 * (size_t)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  tris[3 * (size_t)i + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// PLOC: parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume
// Hierarchy Construction", TVCG 2018).  The radix tree above splits by Morton prefix only, which is what a uniform soup wants but
// wraps big and small triangles of a real scene into the same boxes.  PLOC builds the binary tree bottom up instead: the clusters
// (initially the Morton-sorted leaves) each look `kPlocRadius` neighbours to either side for the partner with the smallest
// merged surface area, mutual choices merge into a new node, the array is compacted, and the round repeats until one cluster is
// left.  The output has the radix tree's array layout (children, boxes, leaf counts in `ranges` as (0, count - 1)), so the
// 8-wide collapse consumes either.  Selected with FH_BVH_BUILDER=ploc.
// ------------------------------------------------------------------------------------------------
constexpr int kPlocRadius = 16;

__global__ void k_ploc_init(int n, const float4* leaf_lo, const float4* leaf_hi, int* cid, float4* clo, float4* chi)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  cid[i] = ~i;
  clo[i] = leaf_lo[i];
  chi[i] = leaf_hi[i];
}

__global__ void k_ploc_nearest(int n, const float4* clo, const float4* chi, int* nn, int radius)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 lo = clo[i], hi = chi[i];
  float best = __builtin_inff();
  int bj = -1;  // the first neighbour is taken unconditionally, so a cluster whose merged areas are all inf / NaN still gets a partner
  const int j0 = i - radius < 0 ? 0 : i - radius, j1 = i + radius > n - 1 ? n - 1 : i + radius;
  for (int j = j0; j <= j1; ++j) {
    if (j == i) continue;
    const float4 l2 = clo[j], h2 = chi[j];
    const float ex = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), ey = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y), ez = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
    const float a = ex * ey + ey * ez + ez * ex;
    if (bj < 0 || a < best) { best = a; bj = j; }  // ties: the lower index (the loop ascends)
  }
  nn[i] = bj;
}

__global__ void k_ploc_merge(int n, const int* nn, int* cid, float4* clo, float4* chi, uint32_t* valid, int2* children, int2* ranges, float4* node_lo, float4* node_hi,
                             uint32_t* node_counter)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = nn[i];
  uint32_t keep = 1u;
  if (j >= 0 && nn[j] == i) {
    if (i < j) {
      const int a = cid[i], b = cid[j];
      const float4 lo = clo[i], hi = chi[i], l2 = clo[j], h2 = chi[j];
      const float4 ulo = make_float4(fminf(lo.x, l2.x), fminf(lo.y, l2.y), fminf(lo.z, l2.z), 0.0f);
      const float4 uhi = make_float4(fmaxf(hi.x, h2.x), fmaxf(hi.y, h2.y), fmaxf(hi.z, h2.z), 0.0f);
      const int id = (int)atomicAdd(node_counter, 1u);
      const int ca = a < 0 ? 1 : ranges[a].y + 1, cb = b < 0 ? 1 : ranges[b].y + 1;
      children[id] = make_int2(a, b);
      ranges[id] = make_int2(0, ca + cb - 1);
      node_lo[id] = ulo;
      node_hi[id] = uhi;
      cid[i] = id;
      clo[i] = ulo;
      chi[i] = uhi;
    } else keep = 0u;
  }
  valid[i] = keep;
}

__global__ void k_ploc_compact(int n, const uint32_t* valid, const uint32_t* offset, const int* cid, const float4* clo, const float4* chi, int* cid_out, float4* clo_out,
                               float4* chi_out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !valid[i]) return;
  const uint32_t o = offset[i];
  cid_out[o] = cid[i];
  clo_out[o] = clo[i];
  chi_out[o] = chi[i];
}

// 1 + the wide node that holds a face into the free lane of the face's record (.z of its last vector; 0: unknown, rays start at the root): the shade kernels, which read that
// vector for the material id, find there where the rays that leave the face start their traversal (fh_trace.h: bottom-up start)
__global__ void k_face_node_to_rec(const uint32_t* face_node, uint32_t n, float4* face_rec)
{
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const uint32_t node = face_node[f];
  face_rec[kFaceRec * (size_t)f + 6].z = __uint_as_float(node == 0xffffffffu ? 0u : node + 1u);
}

// surface-area-heuristic cost of a binary tree: sum over inner nodes of area(node) (the leaves are the same triangles in both builders)
__global__ void k_sah_sum(int n_inner, const float4* node_lo, const float4* node_hi, double* sum)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0.0;
  if (i < n_inner) a = (double)box_area(node_lo[i], node_hi[i]);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
  if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(sum, a);
}

// ------------------------------------------------------------------------------------------------
// Refit of the wide tree.  When only the instance transforms change (Renderer::set_time: the reference rebuilds its IAS and leaves
// the GAS alone, renderer.h:614-640) the topology of the flattened tree is kept: the triangle copies are refreshed from the moved
// face records and the boxes are recomputed bottom up, level by level (levels are contiguous node ranges because the collapse is
// breadth first), with the build's own padding and quantisation rules.  Hits do not depend on the shape of the tree (fh_trace.h), so a
// refitted tree returns the same bits as a rebuilt one; what degrades is its quality, which bvh_build_device watches.
__global__ void k_refresh_tris8(const float4* face_rec, uint32_t n_slots, float4* tris)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const uint32_t f = __float_as_uint(tris[3 * (size_t)i].w);
  if (f == 0xffffffffu) return;  // a slot without a triangle
  const float4 a = face_rec[kFaceRec * (size_t)f], b = face_rec[kFaceRec * (size_t)f + 1], c = face_rec[kFaceRec * (size_t)f + 2];
  tris[3 * (size_t)i] = make_float4(a.x, a.y, a.z, __uint_as_float(f));
  tris[3 * (size_t)i + 1].x = b.x; tris[3 * (size_t)i + 1].y = b.y; tris[3 * (size_t)i + 1].z = b.z;  // .w keeps the alpha flag
  tris[3 * (size_t)i + 2] = make_float4(c.x, c.y, c.z, 0.0f);
}

// full-precision box of every node of a freshly built tree, bottom up (the refit's starting state and its quality reference)
__global__ void k_refit8_level(uint4* nodes, const float4* tris, float4* box, uint32_t begin, uint32_t end, float pad, int requantise)
{
  const uint32_t ni = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (ni >= end) return;
  uint4* nd = nodes + kBvh8NodeVec * (size_t)ni;
  const uint4 n0 = nd[0];
  const uint32_t imask = n0.w & 0xffu, child_base = n0.w >> 8;
  float lo[8][3], hi[8][3];
  bool used[8];
  float nlo[3] = {3e38f, 3e38f, 3e38f}, nhi[3] = {-3e38f, -3e38f, -3e38f};
  for (int sl = 0; sl < 8; ++sl) {
    used[sl] = false;
    if ((imask >> sl) & 1u) {  // inner child: its own box, computed when its level was done (one padding already in it)
      const uint32_t c = child_base + (uint32_t)__popc(imask & ((1u << sl) - 1u));
      const float4 l = box[2 * (size_t)c], h = box[2 * (size_t)c + 1];
      lo[sl][0] = l.x; lo[sl][1] = l.y; lo[sl][2] = l.z; hi[sl][0] = h.x; hi[sl][1] = h.y; hi[sl][2] = h.z;
    } else {  // leaf child: bounds of its triangle, padded like the builder pads leaf boxes
      const size_t t = 3 * (size_t)(8u * ni + (uint32_t)sl);
      const float4 a = tris[t], b = tris[t + 1], c = tris[t + 2];
      if (__float_as_uint(a.w) == 0xffffffffu) continue;  // empty slot
      lo[sl][0] = fminf(a.x, fminf(b.x, c.x)) - pad; lo[sl][1] = fminf(a.y, fminf(b.y, c.y)) - pad; lo[sl][2] = fminf(a.z, fminf(b.z, c.z)) - pad;
      hi[sl][0] = fmaxf(a.x, fmaxf(b.x, c.x)) + pad; hi[sl][1] = fmaxf(a.y, fmaxf(b.y, c.y)) + pad; hi[sl][2] = fmaxf(a.z, fmaxf(b.z, c.z)) + pad;
    }
    used[sl] = true;
    for (int k = 0; k < 3; ++k) { nlo[k] = fminf(nlo[k], lo[sl][k]); nhi[k] = fmaxf(nhi[k], hi[sl][k]); }
  }
  box[2 * (size_t)ni] = make_float4(nlo[0], nlo[1], nlo[2], 0.0f);
  box[2 * (size_t)ni + 1] = make_float4(nhi[0], nhi[1], nhi[2], 0.0f);
  if (!requantise) return;
  float org[3];
  const uint32_t w[3] = {quant_axis(nlo[0], nhi[0], org[0]), quant_axis(nlo[1], nhi[1], org[1]), quant_axis(nlo[2], nhi[2], org[2])};
  const float is[3] = {1.0f / __uint_as_float((w[0] & 0xffu) << 23), 1.0f / __uint_as_float((w[1] & 0xffu) << 23), 1.0f / __uint_as_float((w[2] & 0xffu) << 23)};
  uint32_t q[6][2] = {{~0u, ~0u}, {~0u, ~0u}, {~0u, ~0u}, {0, 0}, {0, 0}, {0, 0}};
  for (int sl = 0; sl < 8; ++sl) {
    if (!used[sl]) continue;
    for (int k = 0; k < 3; ++k) {
      const float vl = fminf(fmaxf(floorf((lo[sl][k] - org[k]) * is[k]), 0.0f), 255.0f), vh = fminf(fmaxf(ceilf((hi[sl][k] - org[k]) * is[k]), 0.0f), 255.0f);
      q[k][sl >> 2] = (q[k][sl >> 2] & ~(0xffu << (8 * (sl & 3)))) | (((uint32_t)vl) << (8 * (sl & 3)));
      q[3 + k][sl >> 2] |= ((uint32_t)vh) << (8 * (sl & 3));
    }
  }
  nd[0] = make_uint4(w[0], w[1], w[2], n0.w);
  nd[1] = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
  nd[2] = make_uint4(q[2][0], q[2][1], q[3][0], q[3][1]);
  nd[3] = make_uint4(q[4][0], q[4][1], q[5][0], q[5][1]);
}

__global__ void k_box_area_sum(uint32_t n, const float4* box, double* sum)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0.0;
  if (i < n) a = (double)box_area(box[2 * (size_t)i], box[2 * (size_t)i + 1]);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
  if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(sum, a);
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

}  // namespace

// boxes of all levels bottom up (requantise = 0: only the full-precision boxes of a freshly built tree), then the sum of the node areas
static int refit_levels(fh_ctx* ctx, float pad, int requantise, double* area)
{
  hipStream_t st = ctx->stream;
  const std::vector<uint32_t>& ls = ctx->bvh8_level_start;
  for (size_t l = ls.size() - 1; l-- > 0;) {
    const uint32_t begin = ls[l], end = ls[l + 1];
    if (end > begin) hipLaunchKernelGGL(k_refit8_level, dim3((end - begin + 127) / 128), dim3(128), 0, st, ctx->d_bvh8_nodes, ctx->d_bvh8_tris, ctx->d_bvh8_box, begin, end, pad, requantise);
  }
  DevBuf<double> sum;
  FH_HIP(sum.alloc(1));
  FH_HIP(hipMemsetAsync(sum.p, 0, 8, st));
  hipLaunchKernelGGL(k_box_area_sum, dim3((ctx->bvh8_n_nodes + 255) / 256), dim3(256), 0, st, ctx->bvh8_n_nodes, ctx->d_bvh8_box, sum.p);
  FH_HIP(hipMemcpyAsync(area, sum.p, 8, hipMemcpyDeviceToHost, st));
  FH_HIP(hipGetLastError());
  FH_HIP(hipStreamSynchronize(st));
  return FH_OK;
}

int bvh_build_device(fh_ctx* ctx)
{
  const auto t_begin = std::chrono::steady_clock::now();
  hipStream_t st = ctx->stream;
  const uint32_t n = ctx->n_faces;
  // ---- instance transforms changed, topology did not: refit (FH_REFIT=0 forces the full rebuild).  A refit whose boxes have grown to
  // more than 1.5x the area the tree had when it was built is discarded for a rebuild: the instances have moved too far for the old topology.
  if (ctx->refit_ok && ctx->use_bvh8 && ctx->d_bvh8_box && n && ctx->bvh8_n_tris == n && !(getenv("FH_REFIT") && getenv("FH_REFIT")[0] == '0')) {
    DevBuf<float4> face_lo, face_hi;
    DevBuf<int> bounds;
    FH_HIP(face_lo.alloc(n)); FH_HIP(face_hi.alloc(n)); FH_HIP(bounds.alloc(6));
    const int init_bounds[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    FH_HIP(hipMemcpyAsync(bounds.p, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_face_bounds, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_face_rec, n, face_lo.p, face_hi.p, bounds.p);
    hipLaunchKernelGGL(k_refresh_tris8, dim3((8u * ctx->bvh8_n_nodes + 255) / 256), dim3(256), 0, st, ctx->d_face_rec, 8u * ctx->bvh8_n_nodes, ctx->d_bvh8_tris);
    int hb[6];
    FH_HIP(hipMemcpyAsync(hb, bounds.p, sizeof hb, hipMemcpyDeviceToHost, st));
    FH_HIP(hipStreamSynchronize(st));
    float maxabs = 0.0f;
    for (int k = 0; k < 6; ++k) maxabs = fmaxf(maxabs, fabsf(order_float(hb[k])));
    const float pad = fmaxf(maxabs, 1e-3f) * (1.0f / 65536.0f);
    double area = 0.0;
    { const int rc = refit_levels(ctx, pad, 1, &area); if (rc) return rc; }
    if (area <= 1.5 * ctx->bvh8_area_built) {
      if (ctx->d_face_node) hipLaunchKernelGGL(k_face_node_to_rec, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_face_node, n, ctx->d_face_rec);  // (k_face_records has just rewritten the records)
      for (int k = 0; k < 3; ++k) { ctx->scene_lo[k] = order_float(hb[k]) - 2.0f * pad; ctx->scene_hi[k] = order_float(hb[3 + k]) + 2.0f * pad; }
      ctx->bvh_valid = true;
      ctx->n_refits++;
      ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
      ctx->stats.bvh_build_ms = ctx->bvh_build_ms;
      if (getenv("FH_DEBUG_BVH")) fprintf(stderr, "[bvh] refit %u: node area %.4f (built %.4f), %.3f ms\n", ctx->n_refits, area, ctx->bvh8_area_built, ctx->bvh_build_ms);
      return FH_OK;
    }
    if (getenv("FH_DEBUG_BVH")) fprintf(stderr, "[bvh] refit discarded: node area %.4f > 1.5 x %.4f, rebuilding\n", area, ctx->bvh8_area_built);
  }
  ctx->refit_ok = false;
  ctx->bu_choice = 0; ctx->bu_toggle = 0; ctx->bu_cost[0] = ctx->bu_cost[1] = ctx->bu_items[0] = ctx->bu_items[1] = 0.0;  // a new tree: where its rays should start is measured again (render.hip)
  ctx->survival_n = 0;  // ... and how long its paths live is not known yet (context.h: survival)
  if (ctx->d_bvh8_box) { (void)hipFree(ctx->d_bvh8_box); ctx->d_bvh8_box = nullptr; }
  ctx->bvh8_level_start.clear();
  if (ctx->d_bvh2_nodes) { (void)hipFree(ctx->d_bvh2_nodes); ctx->d_bvh2_nodes = nullptr; }
  if (ctx->d_bvh2_tris) { (void)hipFree(ctx->d_bvh2_tris); ctx->d_bvh2_tris = nullptr; }
  if (ctx->d_bvh8_nodes) { (void)hipFree(ctx->d_bvh8_nodes); ctx->d_bvh8_nodes = nullptr; }
  if (ctx->d_bvh8_tris) { (void)hipFree(ctx->d_bvh8_tris); ctx->d_bvh8_tris = nullptr; }
  if (ctx->d_bvh8_parent) { (void)hipFree(ctx->d_bvh8_parent); ctx->d_bvh8_parent = nullptr; }
  if (ctx->d_face_node) { (void)hipFree(ctx->d_face_node); ctx->d_face_node = nullptr; }
  ctx->bvh2_n_nodes = ctx->bvh2_n_tris = ctx->bvh8_n_nodes = ctx->bvh8_n_tris = 0;
  ctx->use_bvh8 = false;
  ctx->bvh_valid = false;
  if (n == 0) { ctx->bvh_valid = true; return FH_OK; }

  DevBuf<float4> face_lo, face_hi, node_lo, node_hi, leaf_lo, leaf_hi;
  DevBuf<int> bounds, node_parent, leaf_parent;
  DevBuf<unsigned long long> keys_a, keys_b;
  DevBuf<uint32_t> vals_a, vals_b;
  DevBuf<int2> children, ranges;
  DevBuf<unsigned int> arrive;
  FH_HIP(face_lo.alloc(n)); FH_HIP(face_hi.alloc(n));
  FH_HIP(bounds.alloc(6));
  const int init_bounds[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  FH_HIP(hipMemcpyAsync(bounds.p, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, st));
  const uint32_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_face_bounds, dim3(blocks), dim3(256), 0, st, ctx->d_face_rec, n, face_lo.p, face_hi.p, bounds.p);
  int hb[6];
  FH_HIP(hipMemcpyAsync(hb, bounds.p, sizeof hb, hipMemcpyDeviceToHost, st));
  FH_HIP(hipStreamSynchronize(st));
  float maxabs = 0.0f;
  for (int k = 0; k < 6; ++k) maxabs = fmaxf(maxabs, fabsf(order_float(hb[k])));
  const float pad = fmaxf(maxabs, 1e-3f) * (1.0f / 65536.0f);
  for (int k = 0; k < 3; ++k) { ctx->scene_lo[k] = order_float(hb[k]) - 2.0f * pad; ctx->scene_hi[k] = order_float(hb[3 + k]) + 2.0f * pad; }

  if (n <= kLeafMax2) {
    FH_HIP(hipMalloc((void**)&ctx->d_bvh2_tris, sizeof(float4) * 3ull * n));
    ctx->bvh2_n_tris = n;
    FH_HIP(hipMalloc((void**)&ctx->d_bvh2_nodes, sizeof(float4) * 4));
    FH_HIP(vals_a.alloc(n));
    std::vector<uint32_t> ident(n);
    for (uint32_t i = 0; i < n; ++i) ident[i] = i;
    FH_HIP(hipMemcpyAsync(vals_a.p, ident.data(), 4ull * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_emit2_tiny, dim3(1), dim3(1), 0, st, (int)n, face_lo.p, face_hi.p, pad, ctx->d_bvh2_nodes);
    hipLaunchKernelGGL(k_emit_tris, dim3(blocks), dim3(256), 0, st, ctx->d_face_rec, ctx->d_face_cls, vals_a.p, n, ctx->d_bvh2_tris, (const uint32_t*)nullptr);
    FH_HIP(hipStreamSynchronize(st));
    ctx->bvh2_n_nodes = 1;
    if (n <= kLeafMax8) {
      DevBuf<uint32_t> tri_slot;
      FH_HIP(tri_slot.alloc(n));
      FH_HIP(hipMalloc((void**)&ctx->d_bvh8_nodes, sizeof(uint4) * kBvh8NodeVec));
      FH_HIP(hipMalloc((void**)&ctx->d_bvh8_tris, sizeof(float4) * 3ull * 8ull));
      hipLaunchKernelGGL(k_clear_tris8, dim3(1), dim3(64), 0, st, ctx->d_bvh8_tris, 8u);
      hipLaunchKernelGGL(k_collapse8_tiny, dim3(1), dim3(1), 0, st, (int)n, face_lo.p, face_hi.p, pad, ctx->d_bvh8_nodes, tri_slot.p);
      hipLaunchKernelGGL(k_emit_tris8, dim3(blocks), dim3(256), 0, st, ctx->d_face_rec, ctx->d_face_cls, vals_a.p, tri_slot.p, n, ctx->d_bvh8_tris, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr);
      FH_HIP(hipStreamSynchronize(st));
      ctx->bvh8_n_nodes = 1;
      ctx->bvh8_n_tris = n;
      ctx->use_bvh8 = true;
    }
  } else {
    // ---- references: large triangles enter the build as several clipped boxes (early split clipping, kernels above)
    uint32_t nr = n;
    DevBuf<uint32_t> ref_face_buf, split_count, split_offset;
    DevBuf<float4> ref_lo, ref_hi;
    const float4* box_lo = face_lo.p;
    const float4* box_hi = face_hi.p;
    const uint32_t* ref_face = nullptr;
    bool splitting = n >= 256;
    if (const char* e = getenv("FH_SPLIT")) splitting = splitting && e[0] != '0';
    if (splitting) {
      const float extent = fmaxf(fmaxf(ctx->scene_hi[0] - ctx->scene_lo[0], ctx->scene_hi[1] - ctx->scene_lo[1]), ctx->scene_hi[2] - ctx->scene_lo[2]);
      float thr = extent / 32.0f;
      const float eps = extent * 1e-6f;
      FH_HIP(split_count.alloc(n)); FH_HIP(split_offset.alloc(n));
      size_t scan_bytes = 0;
      FH_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, split_count.p, split_offset.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
      DevBuf<char> scan_tmp;
      FH_HIP(scan_tmp.alloc(scan_bytes));
      uint32_t total = n;
      const double split_budget = 1.5;  // references per face the split may produce at most
      for (int attempt = 0; attempt < 6; ++attempt, thr *= 2.0f) {  // a scene made of large triangles only: coarser cells until the references fit
        hipLaunchKernelGGL(k_split_count, dim3(blocks), dim3(256), 0, st, ctx->d_face_rec, face_lo.p, face_hi.p, n, thr, eps, split_count.p);
        FH_HIP(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, split_count.p, split_offset.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        uint32_t tail[2];
        FH_HIP(hipMemcpyAsync(&tail[0], split_offset.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
        FH_HIP(hipMemcpyAsync(&tail[1], split_count.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
        FH_HIP(hipStreamSynchronize(st));
        total = tail[0] + tail[1];
        if ((unsigned long long)total <= (unsigned long long)((double)n * split_budget) + 4096ull) break;
        total = n;
      }
      if (total > n) {
        nr = total;
        FH_HIP(ref_face_buf.alloc(nr)); FH_HIP(ref_lo.alloc(nr)); FH_HIP(ref_hi.alloc(nr));
        hipLaunchKernelGGL(k_split_emit, dim3(blocks), dim3(256), 0, st, ctx->d_face_rec, face_lo.p, face_hi.p, n, thr, eps, split_count.p, split_offset.p, ref_face_buf.p, ref_lo.p,
                           ref_hi.p);
        box_lo = ref_lo.p; box_hi = ref_hi.p; ref_face = ref_face_buf.p;
      }
    }
    const uint32_t rblocks = (nr + 255) / 256;
    FH_HIP(hipMalloc((void**)&ctx->d_bvh2_tris, sizeof(float4) * 3ull * nr));
    ctx->bvh2_n_tris = nr;
    FH_HIP(keys_a.alloc(nr)); FH_HIP(keys_b.alloc(nr)); FH_HIP(vals_a.alloc(nr)); FH_HIP(vals_b.alloc(nr));
    hipLaunchKernelGGL(k_morton, dim3(rblocks), dim3(256), 0, st, box_lo, box_hi, nr, bounds.p, keys_a.p, vals_a.p);
    size_t temp_bytes = 0;
    FH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, (size_t)nr, 0u, 63u, st));
    DevBuf<char> temp;
    FH_HIP(temp.alloc(temp_bytes));
    FH_HIP(rocprim::radix_sort_pairs(temp.p, temp_bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, (size_t)nr, 0u, 63u, st));
    const uint32_t n_inner = nr - 1;
    FH_HIP(children.alloc(n_inner)); FH_HIP(ranges.alloc(n_inner)); FH_HIP(node_parent.alloc(n_inner)); FH_HIP(leaf_parent.alloc(nr));
    FH_HIP(node_lo.alloc(n_inner)); FH_HIP(node_hi.alloc(n_inner)); FH_HIP(leaf_lo.alloc(nr)); FH_HIP(leaf_hi.alloc(nr));
    FH_HIP(arrive.alloc(n_inner));
    FH_HIP(hipMemsetAsync(arrive.p, 0, 4ull * n_inner, st));
    const uint32_t iblocks = (n_inner + 255) / 256;
    hipLaunchKernelGGL(k_hierarchy, dim3(iblocks), dim3(256), 0, st, keys_b.p, (int)nr, children.p, ranges.p, node_parent.p, leaf_parent.p);
    hipLaunchKernelGGL(k_refit, dim3(rblocks), dim3(256), 0, st, vals_b.p, box_lo, box_hi, (int)nr, children.p, node_parent.p, leaf_parent.p, node_lo.p, node_hi.p, leaf_lo.p,
                       leaf_hi.p, arrive.p);
    FH_HIP(hipMalloc((void**)&ctx->d_bvh2_nodes, sizeof(float4) * 4ull * n_inner));
    hipLaunchKernelGGL(k_emit2, dim3(iblocks), dim3(256), 0, st, (int)n_inner, children.p, ranges.p, node_lo.p, node_hi.p, leaf_lo.p, leaf_hi.p, pad, ctx->d_bvh2_nodes);
    hipLaunchKernelGGL(k_emit_tris, dim3(rblocks), dim3(256), 0, st, ctx->d_face_rec, ctx->d_face_cls, vals_b.p, nr, ctx->d_bvh2_tris, ref_face);
    FH_HIP(hipGetLastError());
    FH_HIP(hipStreamSynchronize(st));
    ctx->bvh2_n_nodes = n_inner;

    // ---- optional: replace the radix tree by a PLOC tree as the input of the collapse (the binary fallback above keeps the radix tree)
    int root_node = 0;
    // FH_BVH_BUILDER = lbvh | ploc | auto (default).  auto: the first build after an upload makes both trees and keeps PLOC when the
    // sum of its inner-node areas is at least 7 % below the radix tree's (non-uniform scenes: -29 % on tools' `city`, 9 % faster
    // frames; -9 % on the Sponza-class glTF, 5 % faster frames); on a uniform soup the two are within 3 % and the radix tree traverses
    // faster (closest-hit rays need the well separated children of a Morton split for the octant order to work), so it stays.  Later builds of the same
    // scene (animation) reuse the choice.
    int mode = ctx->builder_choice;
    if (const char* e = getenv("FH_BVH_BUILDER")) { if (std::strcmp(e, "ploc") == 0) mode = 2; else if (std::strcmp(e, "lbvh") == 0) mode = 1; }
    const bool deciding = mode == 0;
    bool ploc = mode == 2 || deciding;
    bool ploc_failed = false;
    DevBuf<int2> p_children, p_ranges;
    DevBuf<float4> p_node_lo, p_node_hi;
    if (ploc) {
      DevBuf<int> cid_a, cid_b, nn;
      DevBuf<float4> clo_a, chi_a, clo_b, chi_b;
      DevBuf<uint32_t> valid, offset, node_counter;
      FH_HIP(cid_a.alloc(nr)); FH_HIP(cid_b.alloc(nr)); FH_HIP(nn.alloc(nr)); FH_HIP(clo_a.alloc(nr)); FH_HIP(chi_a.alloc(nr)); FH_HIP(clo_b.alloc(nr)); FH_HIP(chi_b.alloc(nr));
      FH_HIP(valid.alloc(nr)); FH_HIP(offset.alloc(nr)); FH_HIP(node_counter.alloc(1));
      FH_HIP(p_children.alloc(n_inner)); FH_HIP(p_ranges.alloc(n_inner)); FH_HIP(p_node_lo.alloc(n_inner)); FH_HIP(p_node_hi.alloc(n_inner));
      FH_HIP(hipMemsetAsync(node_counter.p, 0, 4, st));
      hipLaunchKernelGGL(k_ploc_init, dim3(rblocks), dim3(256), 0, st, (int)nr, leaf_lo.p, leaf_hi.p, cid_a.p, clo_a.p, chi_a.p);
      size_t scan_bytes = 0;
      FH_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, valid.p, offset.p, 0u, (size_t)nr, rocprim::plus<uint32_t>(), st));
      DevBuf<char> scan_tmp;
      FH_HIP(scan_tmp.alloc(scan_bytes));
      const int ploc_radius = kPlocRadius;  // neighbours searched to either side
      int* cid = cid_a.p; int* cid_o = cid_b.p;
      float4 *clo = clo_a.p, *chi = chi_a.p, *clo_o = clo_b.p, *chi_o = chi_b.p;
      uint32_t count = nr;
      for (int round = 0; count > 1 && round < 4096; ++round) {
        const uint32_t b = (count + 255) / 256;
        hipLaunchKernelGGL(k_ploc_nearest, dim3(b), dim3(256), 0, st, (int)count, clo, chi, nn.p, ploc_radius);
        hipLaunchKernelGGL(k_ploc_merge, dim3(b), dim3(256), 0, st, (int)count, nn.p, cid, clo, chi, valid.p, p_children.p, p_ranges.p, p_node_lo.p, p_node_hi.p, node_counter.p);
        FH_HIP(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, valid.p, offset.p, 0u, (size_t)count, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(k_ploc_compact, dim3(b), dim3(256), 0, st, (int)count, valid.p, offset.p, cid, clo, chi, cid_o, clo_o, chi_o);
        uint32_t tail[2] = {0, 0};  // offset and flag of the last cluster -> new count
        FH_HIP(hipMemcpyAsync(&tail[0], offset.p + (count - 1), 4, hipMemcpyDeviceToHost, st));
        FH_HIP(hipMemcpyAsync(&tail[1], valid.p + (count - 1), 4, hipMemcpyDeviceToHost, st));
        FH_HIP(hipStreamSynchronize(st));
        const uint32_t next = tail[0] + tail[1];
        if (next >= count) {  // no mutual pair found (non-finite boxes): in auto mode the radix tree built above is still valid
          if (!deciding) return fail(ctx, FH_E_INVALID, "PLOC made no progress");
          ploc_failed = true;
          break;
        }
        count = next;
        int* t = cid; cid = cid_o; cid_o = t;
        float4* tl = clo; clo = clo_o; clo_o = tl;
        float4* th = chi; chi = chi_o; chi_o = th;
      }
      if (!ploc_failed) {
        FH_HIP(hipMemcpyAsync(&root_node, cid, 4, hipMemcpyDeviceToHost, st));
        FH_HIP(hipStreamSynchronize(st));
        if (root_node < 0 || (uint32_t)root_node >= n_inner) {
          if (!deciding) return fail(ctx, FH_E_INVALID, "PLOC did not end in one inner node");
          ploc_failed = true;
        }
      }
      if (ploc_failed) { ploc = false; ctx->builder_choice = 1; }  // auto mode falls back to the radix tree; an explicit FH_BVH_BUILDER=ploc reported the error above
    }
    if (deciding || getenv("FH_DEBUG_BVH")) {
      DevBuf<double> sums;
      FH_HIP(sums.alloc(2));
      FH_HIP(hipMemsetAsync(sums.p, 0, 16, st));
      hipLaunchKernelGGL(k_sah_sum, dim3(iblocks), dim3(256), 0, st, (int)n_inner, node_lo.p, node_hi.p, sums.p);
      if (ploc) hipLaunchKernelGGL(k_sah_sum, dim3(iblocks), dim3(256), 0, st, (int)n_inner, p_node_lo.p, p_node_hi.p, sums.p + 1);
      double h[2];
      FH_HIP(hipMemcpyAsync(h, sums.p, 16, hipMemcpyDeviceToHost, st));
      FH_HIP(hipStreamSynchronize(st));
      if (deciding && !ploc_failed) {
        ploc = h[1] < 0.93 * h[0];
        ctx->builder_choice = ploc ? 2 : 1;
      }
      if (getenv("FH_DEBUG_BVH")) fprintf(stderr, "[bvh] sum of inner-node areas: radix tree %.4f, PLOC %.4f -> %s\n", h[0], h[1], ploc ? "PLOC" : "radix tree");
    }
    if (!ploc) root_node = 0;  // the radix tree's root is node 0
    const int2* c_children = ploc ? p_children.p : children.p;
    const int2* c_ranges = ploc ? p_ranges.p : ranges.p;
    const float4* c_node_lo = ploc ? p_node_lo.p : node_lo.p;
    const float4* c_node_hi = ploc ? p_node_hi.p : node_hi.p;

    // (SAH refinement of the chosen binary tree by parallel reinsertion, Meister & Bittner 2018, was built in round 5: the summed inner-node area falls by 2-7 %, the node visits per
    // ray do not -- 15.28 -> 15.12 on configs[2], 11.73 -> 11.81 on configs[3].  profiles/README.md r5-1, tools/patches/r6_pruned_switches.patch)

    // ---- collapse to BVH8, breadth first
    DevBuf<Work8> work_a, work_b;
    DevBuf<uint32_t> counters, tri_slot;  // [0] node counter, [1] triangle counter, [2] next-level item count; triangle slot of every sorted leaf
    FH_HIP(work_a.alloc(n_inner)); FH_HIP(work_b.alloc(n_inner)); FH_HIP(counters.alloc(3)); FH_HIP(tri_slot.alloc(nr));
    FH_HIP(hipMalloc((void**)&ctx->d_bvh8_nodes, sizeof(uint4) * (size_t)kBvh8NodeVec * n_inner));
    FH_HIP(hipMalloc((void**)&ctx->d_bvh8_parent, 4ull * n_inner));
    FH_HIP(hipMalloc((void**)&ctx->d_face_node, 4ull * n));
    FH_HIP(hipMemsetAsync(ctx->d_bvh8_parent, 0xff, 4ull * n_inner, st));  // (the root keeps 0xffffffff: no way up)
    FH_HIP(hipMemsetAsync(ctx->d_face_node, 0xff, 4ull * n, st));
    const uint32_t leaf_max8 = 1;  // one triangle per leaf child (the node layout has one triangle slot per child slot): box tests are ~4x cheaper than triangle tests (profiles/README.md)
    uint32_t absorb8 = 1u;  // FH_ABSORB=0: plain largest-child-first collapse
    if (const char* e = getenv("FH_ABSORB")) absorb8 = e[0] != '0' ? 1u : 0u;
    // the cut of least summed wide-node area (k_cut_tables); FH_COLLAPSE=greedy: the largest-child-first rule above
    bool optimal_cut = leaf_max8 == 1;
    if (const char* e = getenv("FH_COLLAPSE")) optimal_cut = optimal_cut && std::strcmp(e, "greedy") != 0;
    DevBuf<float> cut_cost_tab;
    DevBuf<uint2> cut_decision;
    if (optimal_cut) {
      DevBuf<int> c_node_parent, c_leaf_parent;
      FH_HIP(c_node_parent.alloc(n_inner)); FH_HIP(c_leaf_parent.alloc(nr)); FH_HIP(cut_cost_tab.alloc(7ull * n_inner)); FH_HIP(cut_decision.alloc(n_inner));
      FH_HIP(hipMemsetAsync(c_node_parent.p, 0xff, 4ull * n_inner, st));  // the root keeps -1
      FH_HIP(hipMemsetAsync(arrive.p, 0, 4ull * n_inner, st));
      hipLaunchKernelGGL(k_tree_parents, dim3(iblocks), dim3(256), 0, st, (int)n_inner, c_children, c_node_parent.p, c_leaf_parent.p);
      hipLaunchKernelGGL(k_cut_tables, dim3(rblocks), dim3(256), 0, st, (int)nr, c_children, c_node_parent.p, c_leaf_parent.p, c_node_lo, c_node_hi, arrive.p, cut_cost_tab.p,
                         cut_decision.p);
      FH_HIP(hipGetLastError());
      FH_HIP(hipStreamSynchronize(st));  // (the parent arrays go out of scope here)
      if (getenv("FH_DEBUG_BVH")) {
        float c_root = 0.0f;
        FH_HIP(hipMemcpy(&c_root, cut_cost_tab.p + 7ull * (size_t)root_node, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[bvh] least summed area of the wide nodes (cut tables, root): %.4f\n", c_root);
      }
    }
    const Work8 root{root_node, 0u};
    const uint32_t init_counters[3] = {1u, 0u, 0u};
    FH_HIP(hipMemcpyAsync(work_a.p, &root, sizeof root, hipMemcpyHostToDevice, st));
    FH_HIP(hipMemcpyAsync(counters.p, init_counters, sizeof init_counters, hipMemcpyHostToDevice, st));
    uint32_t level_count = 1;
    Work8* cur = work_a.p;
    Work8* nxt = work_b.p;
    uint32_t levels = 0;
    std::vector<uint32_t> level_start{0u};
    for (int level = 0; level < 64 && level_count > 0; ++level) {
      ++levels;
      level_start.push_back(level_start.back() + level_count);
      FH_HIP(hipMemsetAsync(counters.p + 2, 0, 4, st));
      hipLaunchKernelGGL(k_collapse8, dim3((level_count + 63) / 64), dim3(64), 0, st, cur, level_count, c_children, c_ranges, c_node_lo, c_node_hi, leaf_lo.p, leaf_hi.p, pad,
                         leaf_max8, absorb8, ctx->d_bvh8_nodes, counters.p, counters.p + 1, tri_slot.p, nxt, counters.p + 2, optimal_cut ? cut_decision.p : (const uint2*)nullptr, ctx->d_bvh8_parent);
      FH_HIP(hipMemcpyAsync(&level_count, counters.p + 2, 4, hipMemcpyDeviceToHost, st));
      FH_HIP(hipStreamSynchronize(st));
      Work8* t = cur; cur = nxt; nxt = t;
    }
    uint32_t final_counters[2] = {0, 0};
    FH_HIP(hipMemcpyAsync(final_counters, counters.p, 8, hipMemcpyDeviceToHost, st));
    FH_HIP(hipStreamSynchronize(st));
    if (final_counters[1] != nr) return fail(ctx, FH_E_INVALID, "BVH8 collapse lost triangles");
    if (levels > (uint32_t)kBvh8Stack) return fail(ctx, FH_E_UNSUPPORTED, "BVH8 deeper than the traversal stack (48 levels)");
    if (final_counters[0] >= (1u << 24)) return fail(ctx, FH_E_UNSUPPORTED, "BVH8 with 2^24 or more nodes (the traversal stack keeps node indices in 24 bits)");
    if (final_counters[0] >= kCoopMaxTris / 8u) return fail(ctx, FH_E_UNSUPPORTED, "BVH8 with 2^23 or more nodes (the cooperative triangle queue keeps triangle slots in 26 bits)");
    ctx->bvh8_depth = levels;
    const uint32_t n_slots = 8u * final_counters[0];  // one triangle slot per child slot
    FH_HIP(hipMalloc((void**)&ctx->d_bvh8_tris, sizeof(float4) * 3ull * n_slots));
    hipLaunchKernelGGL(k_clear_tris8, dim3((n_slots + 255) / 256), dim3(256), 0, st, ctx->d_bvh8_tris, n_slots);
    hipLaunchKernelGGL(k_emit_tris8, dim3(rblocks), dim3(256), 0, st, ctx->d_face_rec, ctx->d_face_cls, vals_b.p, tri_slot.p, nr, ctx->d_bvh8_tris, ref_face, ctx->d_face_node, ref_face ? split_count.p : (const uint32_t*)nullptr);
    hipLaunchKernelGGL(k_face_node_to_rec, dim3(blocks), dim3(256), 0, st, ctx->d_face_node, n, ctx->d_face_rec);
    FH_HIP(hipGetLastError());
    FH_HIP(hipStreamSynchronize(st));
    ctx->bvh8_n_nodes = final_counters[0];
    ctx->bvh8_n_tris = nr;
    ctx->use_bvh8 = true;
    // what a later refit needs: the level ranges, the full-precision node boxes and the area the tree has now.  Split references carry
    // clipped boxes that a moved triangle no longer has, so scenes that use them are rebuilt instead.
    if (nr == n && level_start.back() == final_counters[0]) {
      ctx->bvh8_level_start = level_start;
      FH_HIP(hipMalloc((void**)&ctx->d_bvh8_box, sizeof(float4) * 2ull * final_counters[0]));
      const int rc = refit_levels(ctx, pad, 0, &ctx->bvh8_area_built);
      if (rc) return rc;
      ctx->refit_ok = true;
      ctx->n_refits = 0;
    }
  }
  ctx->bvh_valid = true;
  ctx->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  ctx->stats.bvh_build_ms = ctx->bvh_build_ms;
  if (getenv("FH_BVH2")) ctx->use_bvh8 = false;  // developer switch: traverse the binary layout instead of the wide one
  ctx->stats.bvh_nodes = ctx->use_bvh8 ? ctx->bvh8_n_nodes : ctx->bvh2_n_nodes;
  ctx->stats.bvh_node_bytes = ctx->use_bvh8 ? 16ull * kBvh8NodeVec * ctx->bvh8_n_nodes : 64ull * ctx->bvh2_n_nodes;
  ctx->stats.bvh_tri_bytes = ctx->use_bvh8 ? 48ull * 8ull * ctx->bvh8_n_nodes : 48ull * ctx->bvh2_n_tris;  // (wide tree: eight triangle slots per node)
  ctx->stats.bvh_depth = ctx->use_bvh8 ? ctx->bvh8_depth : 0;
  return FH_OK;
}

}  // namespace fh
