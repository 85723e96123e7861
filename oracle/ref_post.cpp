// ref_post.cpp -- TEST INFRASTRUCTURE: a thin driver around the REFERENCE's own tone-map helper chain, compiled for the host
// from where it lies under /root/reference (never copied into this repository):
//   fredholm/kernels/include/kernels/post-process.h:13-124   rgb_to_luminance, linear_to_srgb, aces_tone_mapping, step, smoothstep,
//                                                            uchimura, compute_EV100, convert_EV100_to_exposure
// It is a translation unit of its own because math.cu defines rgb_to_luminance too.  The kernels that call these helpers
// (kernels/src/post-process.cu) include cwl/util.h -> the CUDA driver API and use <<<>>> launches: unbuildable here; what
// is pinned is the arithmetic of every per-pixel helper and, below, the helper composition of tone_mapping_kernel's tail
// (post-process.cu:139-152: exposure -> uchimura -> linear_to_srgb) called in the reference's order.
// Built by oracle/Makefile into oracle/_ref/libref_lut_math_post.so.
#include <math.h>

#include "kernels/post-process.h"

extern "C" {

void ref_post_luminance(int n, const float* rgb3, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = rgb_to_luminance(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
}

void ref_linear_to_srgb(int n, const float* rgb3, float* out3)
{
  for (int i = 0; i < n; ++i) {
    const float3 r = linear_to_srgb(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}

void ref_uchimura(int n, const float* rgb3, float* out3)
{
  for (int i = 0; i < n; ++i) {
    const float3 r = uchimura(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}

void ref_aces(int n, const float* rgb3, float* out3)
{
  for (int i = 0; i < n; ++i) {
    const float3 r = aces_tone_mapping(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
  }
}

void ref_smoothstep(int n, float edge0, float edge1, const float* x, float* out)
{
  for (int i = 0; i < n; ++i) out[i] = smoothstep(edge0, edge1, x[i]);
}

// compute_EV100(aperture, shutter, ISO) and convert_EV100_to_exposure of it
void ref_exposure(int n, const float* aperture, const float* shutter, const float* iso, float* ev100, float* exposure)
{
  for (int i = 0; i < n; ++i) {
    ev100[i] = compute_EV100(aperture[i], shutter[i], iso[i]);
    exposure[i] = convert_EV100_to_exposure(ev100[i]);
  }
}

// the per-pixel tail of tone_mapping_kernel (post-process.cu:139-152) on colours already fetched
void ref_tone_map_tail(int n, float iso, const float* rgb3, float* out3)
{
  const float EV100 = compute_EV100(1.0f, 1.0f, iso);
  const float exposure = convert_EV100_to_exposure(EV100);
  for (int i = 0; i < n; ++i) {
    float3 color = make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]);
    color *= exposure;
    color = uchimura(color);
    color = linear_to_srgb(color);
    out3[3 * i] = color.x; out3[3 * i + 1] = color.y; out3[3 * i + 2] = color.z;
  }
}

}  // extern "C"
