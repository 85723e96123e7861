// oracle/ovec.h -- TEST INFRASTRUCTURE (CPU checker), never linked into the product.
//
// Minimal fp32 vector algebra with the evaluation order of the reference's vector helper
// (externals/sutil/sutil/vec_math.h): dot = x*x' + y*y' + z*z' left to right (:549),
// v / s == v * (1/s) (:498-502), normalize(v) == v * (1/sqrt(dot(v,v))) (:568-572),
// clamp(f,a,b) == fmaxf(a, fminf(f,b)) (:115-119, so a NaN clamps to b... then to max(a,b)),
// lerp(a,b,t) == a + t*(b-a) (:515-519).  Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct U3 { uint32_t x, y, z; };

inline V2 v2(float x, float y) { return {x, y}; }
inline V2 v2(float s) { return {s, s}; }
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 v3(float s) { return {s, s, s}; }
inline V3 v3(const V4& a) { return {a.x, a.y, a.z}; }
inline V4 v4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline V4 v4(const V3& a, float w) { return {a.x, a.y, a.z, w}; }

inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator-(V2 a, float b) { return {a.x - b, a.y - b}; }
inline V2 operator*(float s, V2 a) { return {s * a.x, s * a.y}; }
inline V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator+(V3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline V3 operator+(float b, V3 a) { return {b + a.x, b + a.y, b + a.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline V3 operator-(float b, V3 a) { return {b - a.x, b - a.y, b - a.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator/(V3 a, float s) { const float inv = 1.0f / s; return a * inv; }
inline V3 operator/(float s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
inline V3& operator/=(V3& a, float s) { const float inv = 1.0f / s; a = a * inv; return a; }

inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator*(float s, V4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
inline V4 operator/(V4 a, float s) { const float inv = 1.0f / s; return {a.x * inv, a.y * inv, a.z * inv, a.w * inv}; }
inline V4& operator+=(V4& a, V4 b) { a = a + b; return a; }

inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(V4 a, V4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { const float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
inline int clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }  // max(a, min(f, b))
inline uint32_t clampu(uint32_t f, uint32_t a, uint32_t b) { uint32_t m = f < b ? f : b; return a > m ? a : m; }
inline V3 clamp3(V3 v, V3 a, V3 b) { return {clampf(v.x, a.x, b.x), clampf(v.y, a.y, b.y), clampf(v.z, a.z, b.z)}; }
inline V2 clamp2(V2 v, V2 a, V2 b) { return {clampf(v.x, a.x, b.x), clampf(v.y, a.y, b.y)}; }
inline V3 lerp3(V3 a, V3 b, float t) { return a + t * (b - a); }
inline V3 fmax3(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
inline V3 sqrt3(V3 a) { return {sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
inline bool anynan(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
inline bool anyinf(V3 a) { return std::isinf(a.x) || std::isinf(a.y) || std::isinf(a.z); }

inline int f2i(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float i2f(int i) { float f; std::memcpy(&f, &i, 4); return f; }

// 3x4 row matrix (reference Matrix3x4, shared.h:11-50)
struct M34 { V4 r[3]; };
inline V3 xform_point(const M34& m, V3 p) { V4 v = {p.x, p.y, p.z, 1.0f}; return {dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)}; }
inline V3 xform_dir(const M34& m, V3 d) { V4 v = {d.x, d.y, d.z, 0.0f}; return {dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)}; }
// multiply by the transpose of the 3x3 part (shared.h:42-50)
inline V3 xform_normal(const M34& m, V3 n)
{
  V4 c0 = {m.r[0].x, m.r[1].x, m.r[2].x, 0.0f}, c1 = {m.r[0].y, m.r[1].y, m.r[2].y, 0.0f}, c2 = {m.r[0].z, m.r[1].z, m.r[2].z, 0.0f};
  V4 t = {n.x, n.y, n.z, 0.0f};
  return {dot(c0, t), dot(c1, t), dot(c2, t)};
}

constexpr float kPi = 3.14159265358979323846f;  // M_PIf

}  // namespace orc
