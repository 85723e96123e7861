// fredholm/types.h -- the handful of CUDA vector types the reference's host API is written against
// (float2/float3/float4/uint3 and their make_* constructors, from <cuda_runtime.h> via sutil/vec_math.h).
// When a HIP translation unit already provides them (hip/hip_vector_types.h) nothing is defined here.
#pragma once
#include <cstdint>

#if !defined(__HIPCC__) && !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(__VECTOR_TYPES_H__)
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct uint2 { unsigned int x, y; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float s) { return {s, s, s}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint3 make_uint3(unsigned int x, unsigned int y, unsigned int z) { return {x, y, z}; }
#endif
using uint = unsigned int;
