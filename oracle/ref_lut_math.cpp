// ref_lut_math.cpp -- TEST INFRASTRUCTURE: a thin driver around the REFERENCE's own device functions that need nothing but
// CUDA's vector types, compiled for the host from where they lie under /root/reference (never copied into this repository):
//   fredholm/modules/lut.cu   directional-albedo tables and their bilinear / trilinear fetchers   (:5-955 data, :957-1081 fetch)
//   fredholm/modules/math.cu  orthonormal_basis, world_to_local, local_to_world, rgb_to_luminance, cartesian_to_spherical (:7-35, :90-118)
// Both include only "sutil/vec_math.h" (vendored in the reference under externals/sutil), which needs <vector_types.h> /
// <vector_functions.h>: the real CUDA headers ship inside the triton wheel of this image, so no stand-in is written.
// Built by oracle/Makefile into oracle/_ref/libref_lut_math_post.so (git-ignored, travels to the GPU box with the snapshot).
// tests/golden/gen_ref_golden.py runs it to produce the committed fixtures.
#include <math.h>

#include "lut.cu"
#include "math.cu"

extern "C" {

// compute_directional_albedo_reflection (lut.cu:985-992) for n (w.y, roughness, F0) triples
void ref_albedo_reflection(int n, const float* wy, const float* roughness, const float* F0, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = compute_directional_albedo_reflection(make_float3(0.0f, wy[i], 0.0f), roughness[i], F0[i]);
}

// compute_directional_albedo_reflection_ior1 (lut.cu:1038-1045)
void ref_albedo_reflection_ior1(int n, const float* wy, const float* roughness, const float* eta, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = compute_directional_albedo_reflection_ior1(make_float3(0.0f, wy[i], 0.0f), roughness[i], eta[i]);
}

// compute_directional_albedo_sheen (lut.cu:1075-1081)
void ref_albedo_sheen(int n, const float* wy, const float* roughness, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = compute_directional_albedo_sheen(make_float3(0.0f, wy[i], 0.0f), roughness[i]);
}

// raw table entries (lut.cu:957-963, :994-1003, :1047-1053), clamped indices
void ref_lut_entries(float* reflection512, float* sheen256)
{
  for (int j = 0; j < REFLECTION_LUT_SIZE; ++j)
    for (int i = 0; i < REFLECTION_LUT_SIZE; ++i) {
      const float2 t = fetch_reflection_lut_idx(i, j);
      reflection512[2 * i + 2 * REFLECTION_LUT_SIZE * j] = t.x;
      reflection512[2 * i + 2 * REFLECTION_LUT_SIZE * j + 1] = t.y;
    }
  for (int j = 0; j < SHEEN_LUT_SIZE; ++j)
    for (int i = 0; i < SHEEN_LUT_SIZE; ++i) sheen256[i + SHEEN_LUT_SIZE * j] = fetch_sheen_lut_idx(i, j);
}

// orthonormal_basis (math.cu:7-17): n normals (xyz) -> tangent, bitangent
void ref_orthonormal_basis(int n, const float* nrm3, float* tangent3, float* bitangent3)
{
  for (int i = 0; i < n; ++i) {
    float3 t, b;
    orthonormal_basis(make_float3(nrm3[3 * i], nrm3[3 * i + 1], nrm3[3 * i + 2]), t, b);
    tangent3[3 * i] = t.x; tangent3[3 * i + 1] = t.y; tangent3[3 * i + 2] = t.z;
    bitangent3[3 * i] = b.x; bitangent3[3 * i + 1] = b.y; bitangent3[3 * i + 2] = b.z;
  }
}

// world_to_local / local_to_world (math.cu:19-35) in the frame (t, n, b)
void ref_world_to_local(int n, const float* v3, const float* t3, const float* n3, const float* b3, float* out3)
{
  for (int i = 0; i < n; ++i) {
    const float3 r = world_to_local(make_float3(v3[3 * i], v3[3 * i + 1], v3[3 * i + 2]), make_float3(t3[3 * i], t3[3 * i + 1], t3[3 * i + 2]),
                                    make_float3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]), make_float3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}
void ref_local_to_world(int n, const float* v3, const float* t3, const float* n3, const float* b3, float* out3)
{
  for (int i = 0; i < n; ++i) {
    const float3 r = local_to_world(make_float3(v3[3 * i], v3[3 * i + 1], v3[3 * i + 2]), make_float3(t3[3 * i], t3[3 * i + 1], t3[3 * i + 2]),
                                    make_float3(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]), make_float3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}

// cartesian_to_spherical (math.cu:111-118): (theta, phi); host libm acosf / atan2f
void ref_cartesian_to_spherical(int n, const float* w3, float* out2)
{
  for (int i = 0; i < n; ++i) {
    const float2 r = cartesian_to_spherical(make_float3(w3[3 * i], w3[3 * i + 1], w3[3 * i + 2]));
    out2[2 * i] = r.x; out2[2 * i + 1] = r.y;
  }
}

// rgb_to_luminance (math.cu:90-93)
void ref_rgb_to_luminance(int n, const float* rgb3, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = rgb_to_luminance(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
}

}  // extern "C"
