// fh_vec.h -- fp32 vector algebra for the HIP kernels (and their host-side setup code).
//
// Evaluation order follows the vector helper the reference integrator is written against
// (externals/sutil/sutil/vec_math.h): dot is summed left to right (:549), vector / scalar
// multiplies by the reciprocal (:498-502), normalize multiplies by 1/sqrt(dot) (:568-572),
// clamp is fmaxf(a, fminf(f, b)) (:115-119) so a NaN clamps to the upper bound, and
// lerp(a,b,t) = a + t*(b-a) (:515-519).  Everything is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/fh_elementary.h"

#define FH_HD __host__ __device__ __forceinline__
#define FH_D __device__ __forceinline__

namespace fh {

struct f2 { float x, y; };
struct f3 { float x, y, z; };

FH_HD f2 mk2(float x, float y) { return {x, y}; }
FH_HD f3 mk3(float x, float y, float z) { return {x, y, z}; }
FH_HD f3 mk3(float s) { return {s, s, s}; }
FH_HD f3 mk3(const float4& a) { return {a.x, a.y, a.z}; }
FH_HD float4 mk4(f3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

FH_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
FH_HD f2 operator-(f2 a, float b) { return {a.x - b, a.y - b}; }
FH_HD f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }

FH_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
FH_HD f3 operator+(f3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
FH_HD f3 operator+(float b, f3 a) { return {b + a.x, b + a.y, b + a.z}; }
FH_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
FH_HD f3 operator-(f3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
FH_HD f3 operator-(float b, f3 a) { return {b - a.x, b - a.y, b - a.z}; }
FH_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
FH_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
FH_HD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
FH_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
FH_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
FH_HD f3 operator/(f3 a, float s) { const float inv = 1.0f / s; return a * inv; }
FH_HD f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
FH_HD f3& operator*=(f3& a, f3 b) { a = a * b; return a; }

// correctly rounded square root (include/fh_elementary.h: the short device sequence, bit-identical to sqrtf)
FH_HD float sqrt_cr(float x) { return fhe_sqrt(x); }

FH_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FH_HD f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
FH_HD float length(f3 a) { return sqrt_cr(dot(a, a)); }
FH_HD f3 normalize(f3 a) { const float inv = 1.0f / sqrt_cr(dot(a, a)); return a * inv; }
FH_HD float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
FH_HD int clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }
FH_HD f3 clamp01(f3 v) { return {clampf(v.x, 0.0f, 1.0f), clampf(v.y, 0.0f, 1.0f), clampf(v.z, 0.0f, 1.0f)}; }
FH_HD f3 max3(f3 a, f3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
FH_HD f3 sqrt3(f3 a) { return {sqrt_cr(a.x), sqrt_cr(a.y), sqrt_cr(a.z)}; }
FH_HD bool bad1(float v) { return isnan(v) || isinf(v); }
FH_HD bool bad3(f3 v) { return bad1(v.x) || bad1(v.y) || bad1(v.z); }
FH_HD float lum(f3 c) { return dot(c, mk3(0.2126729f, 0.7151522f, 0.0721750f)); }

// rows of a 3x4 affine matrix (reference Matrix3x4, shared.h:11-50)
struct m34 { float4 r[3]; };
FH_HD float dot4(const float4& a, float x, float y, float z, float w) { return a.x * x + a.y * y + a.z * z + a.w * w; }
FH_HD f3 xform_point(const m34& m, f3 p) { return {dot4(m.r[0], p.x, p.y, p.z, 1.0f), dot4(m.r[1], p.x, p.y, p.z, 1.0f), dot4(m.r[2], p.x, p.y, p.z, 1.0f)}; }
FH_HD f3 xform_dir(const m34& m, f3 d) { return {dot4(m.r[0], d.x, d.y, d.z, 0.0f), dot4(m.r[1], d.x, d.y, d.z, 0.0f), dot4(m.r[2], d.x, d.y, d.z, 0.0f)}; }
FH_HD f3 xform_normal(const m34& m, f3 n)  // transpose of the 3x3 block, shared.h:42-50
{
  return {m.r[0].x * n.x + m.r[1].x * n.y + m.r[2].x * n.z + 0.0f * 0.0f, m.r[0].y * n.x + m.r[1].y * n.y + m.r[2].y * n.z + 0.0f * 0.0f,
          m.r[0].z * n.x + m.r[1].z * n.y + m.r[2].z * n.z + 0.0f * 0.0f};
}

constexpr float kPi = 3.14159265358979323846f;

// Duff et al. branchless orthonormal basis (math.cu:7-17)
FH_HD void onb(f3 n, f3& t, f3& b)
{
  const float sign = copysignf(1.0f, n.z);
  const float a = -1.0f / (sign + n.z);
  const float bb = n.x * n.y * a;
  t = mk3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
  b = mk3(bb, sign + n.y * n.y * a, -n.y);
}
FH_HD f3 to_local(f3 v, f3 t, f3 n, f3 b) { return mk3(dot(v, t), dot(v, n), dot(v, b)); }
FH_HD f3 to_world(f3 v, f3 t, f3 n, f3 b)
{
  return mk3(v.x * t.x + v.y * n.x + v.z * b.x, v.x * t.y + v.y * n.y + v.z * b.y, v.x * t.z + v.y * n.z + v.z * b.z);
}

// RT-Gems ch.6 origin offset (pt.cu:402-416)
FH_D f3 offset_origin(f3 p, f3 n)
{
  const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
  const int ox = (int)(int_scale * n.x), oy = (int)(int_scale * n.y), oz = (int)(int_scale * n.z);
  const float px = __int_as_float(__float_as_int(p.x) + ((p.x < 0) ? -ox : ox));
  const float py = __int_as_float(__float_as_int(p.y) + ((p.y < 0) ? -oy : oy));
  const float pz = __int_as_float(__float_as_int(p.z) + ((p.z < 0) ? -oz : oz));
  return mk3(fabsf(p.x) < origin ? p.x + float_scale * n.x : px, fabsf(p.y) < origin ? p.y + float_scale * n.y : py, fabsf(p.z) < origin ? p.z + float_scale * n.z : pz);
}

}  // namespace fh
