/* fredholm_hip_test.h -- known-answer entry points of libfredholm_hip.so for the parity tests.
 *
 * NOT part of the interface that mirrors the reference (include/fredholm_hip.h): each function evaluates, on the device, one piece of the code the render
 * kernels are made of (hashes, samplers, warps, the BSDF, the sky, the camera, the texture unit ...) for a batch of inputs, so that tests/ can compare it bit
 * for bit with the CPU checker (oracle/) and with the parts of the reference that build here (oracle/_ref).  An application never calls them. */
#ifndef FREDHOLM_HIP_TEST_H
#define FREDHOLM_HIP_TEST_H
#include "fredholm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* kind 0: xxhash32(a) ; 1: xxhash32(a,b,c) ; 2: xxhash32(a,b,c,d) ; 3: cmj_permute(a,b,c); in: uint32[4] per item */
int fh_kat_hash(fh_ctx* ctx, int kind, uint32_t n, const uint32_t* in4, uint32_t* out);
/* CMJ 2-D draws: in = (n_spp, image_idx, slot, seed) per item, out 2 floats */
int fh_kat_cmj(fh_ctx* ctx, uint32_t n, const uint32_t* in4, float* out2);
/* Owen-Sobol draws: in = (index32, dimension, seed_hash, unused) per item, out 1 float */
int fh_kat_sobol(fh_ctx* ctx, uint32_t n, const uint32_t* in4, float* out);
/* elementary functions of include/fh_elementary.h evaluated on the device; fn as in the checker */
int fh_kat_elementary(fh_ctx* ctx, int fn, uint32_t n, const float* x, const float* y, float* out);
/* warps: kind 0 disk, 1 cosine hemisphere, 2 triangle, 3 vndf(wo, alpha) */
int fh_kat_warp(fh_ctx* ctx, int kind, uint32_t n, const float* u2, const float* wo3, const float* alpha2, float* out);
/* BSDF: 18 floats per item {eval.rgb, pdf, sample.wi, sample.f, sample.pdf, lobe weights-as-pmf[7]} */
int fh_kat_bsdf(fh_ctx* ctx, const fh_material* material, int entering, uint32_t lobes_mask, uint32_t n, const float* wo3, const float* wi3, const float* u1, const float* u2, float* out18);
/* the same with the interface's relative index of refraction GIVEN (eta > 0) instead of the constructor's 1.5 (bsdf.cu:16-18), entering = true: the lobe classes take it as
   an argument (bxdf.cu:433-442, :620-627) and the reference's REFLECTION_IOR1_LUT (lut.cu:94-916) tabulates the dielectric reflection lobe over eta in (0, 1) */
int fh_kat_bsdf_ior(fh_ctx* ctx, const fh_material* material, float eta, uint32_t lobes_mask, uint32_t n, const float* wo3, const float* wi3, const float* u1, const float* u2, float* out18);
int fh_kat_sky(fh_ctx* ctx, uint32_t n, const float* dirs3, float* out3);          /* uses the context's Hosek state */
int fh_kat_hosek_state(fh_ctx* ctx, float* out30);                                  /* cooked cfg[3][9] + rad[3] */
int fh_kat_camera(fh_ctx* ctx, const fh_camera* cam, uint32_t width, uint32_t height, uint32_t seed, uint32_t n, const uint32_t* pixel_idx, const uint32_t* n_spp, float* out6);
int fh_kat_offset_origin(fh_ctx* ctx, uint32_t n, const float* p3, const float* n3, float* out3);
/* small math blocks that have a reference-built counterpart (oracle/_ref/libref_lut_math_post.so); floats in / out per item:
   ALBEDO_REFLECTION (w.y, roughness, F0) -> 1  lut.cu:985-992     ALBEDO_SHEEN (w.y, roughness) -> 1  lut.cu:1075-1081
   ONB n.xyz -> tangent.xyz bitangent.xyz  math.cu:7-17            TO_LOCAL / TO_WORLD (v, t, n, b) -> 3  math.cu:19-35
   SPHERICAL w.xyz -> (theta, phi)  math.cu:111-118                LUMINANCE rgb -> 1  math.cu:90-93
   UCHIMURA rgb -> 3  post-process.h:78-111                        LINEAR_TO_SRGB rgb -> 3  post-process.h:19-29
   EXPOSURE (aperture, shutter, ISO) -> (EV100, exposure)  post-process.h:114-125
   TONE_MAP_TAIL (r, g, b, ISO) -> 3  post-process.cu:139-152      POST_LUMINANCE rgb -> 1  post-process.h:13-16 */
enum { FH_MATH_ALBEDO_REFLECTION = 0, FH_MATH_ALBEDO_SHEEN, FH_MATH_ONB, FH_MATH_TO_LOCAL, FH_MATH_TO_WORLD, FH_MATH_SPHERICAL, FH_MATH_LUMINANCE,
       FH_MATH_UCHIMURA, FH_MATH_LINEAR_TO_SRGB, FH_MATH_EXPOSURE, FH_MATH_TONE_MAP_TAIL, FH_MATH_POST_LUMINANCE, FH_MATH_COUNT };
int fh_kat_math(fh_ctx* ctx, int kind, uint32_t n, const float* in, float* out);
/* tex2D<float4>() of the software texture unit (include/fh_texture_unit.h: cwl/texture.h:35-47 semantics) evaluated on the device for n (u, v)
   pairs on an RGBA8 texture (rgba8, optionally sRGB) or a float4 texture (rgba32f); exactly one of the two texel pointers is non-NULL */
/* the device's short correctly rounded square root (include/fh_elementary.h: fhe_sqrt) against the compiler's IEEE sqrtf over all 2^32 float bit patterns
 * (number of disagreeing inputs; 0 expected), plus its results on `n_sample` given inputs for a comparison with the host's sqrtf */
int fh_kat_sqrt(fh_ctx* ctx, unsigned long long* mismatches_over_all_inputs, uint32_t n_sample, const float* sample_in, float* sample_out);
int fh_kat_tex2d(fh_ctx* ctx, const uint8_t* rgba8, const float* rgba32f, uint32_t width, uint32_t height, int srgb, uint32_t n, const float* uv2, float* out4);
/* the class byte of every face of the uploaded scene (n = faces): bits 0-4 shading class, 0x20 a cut-out face whose any-hit test can never pass (no ray hits it),
   0x40 the any-hit test runs for candidate hits on this face, 0x80 emissive.  A face of a material whose textures can cut with neither 0x20 nor 0x40 was
   classified "always passes" from the texels it can address (capi.hip: footprint_class); the parity tests check both classes by brute force. */
int fh_kat_face_classes(fh_ctx* ctx, uint8_t* out, uint32_t n);
/* the any-hit record of every face (n_faces x 32 words: texture coordinates, flags, texture pointers, then 16 words of opacity micromap: two bits per cell, cell = 16 * floor(16 v) + floor(16 u));
   all zero for scenes without cut-outs */
int fh_kat_alpha_records(fh_ctx* ctx, uint32_t* out, uint32_t n_faces);
/* where the scene's first-hit rays start their traversal (render.hip, "where this pass's first-hit rays start"): out[0] = 0 still probing, 1 the root, 2 the wide node of
   the face they leave; out[1..2] = shaded paths the probing passes have counted from the root / from the face, out[3..4] = their test cycles by the issue model's weights */
int fh_kat_ray_start(fh_ctx* ctx, double out[5]);

#ifdef __cplusplus
}
#endif
#endif
