// fredholm/denoiser.h -- drop-in for the reference's Denoiser (fredholm/include/fredholm/denoiser.h:14-146), which wraps the proprietary OptiX
// AI denoiser (HDR model, albedo + normal guide layers, optional 2x upscaling model).  That network has no counterpart here; the class keeps the
// constructor / denoise() / wait_for_completion() surface the applications call (app/controller.cpp:70-78,232-236, app/rtcamp8.cpp:120-128,191-196)
// and runs the library's guided filter in its place: an edge-avoiding a-trous wavelet filter (Dammertz et al. 2010) on albedo-demodulated
// radiance, steered by the same normal and albedo layers (fh_denoise).  Output size follows the reference: 2 x width by 2 x height when `upscale`.
#pragma once
#include <cstdint>

#include "../cwl/util.h"
#include "types.h"

namespace fredholm
{
class Denoiser
{
 public:
  Denoiser(fh_ctx* context, uint32_t width, uint32_t height, const float4* d_beauty, const float4* d_normal, const float4* d_albedo, const float4* d_denoised, bool upscale = false)
      : m_context(context), m_width(width), m_height(height), m_d_beauty(d_beauty), m_d_normal(d_normal), m_d_albedo(d_albedo), m_d_denoised(const_cast<float4*>(d_denoised)), m_upscale(upscale)
  {
  }
  void denoise()
  {
    fh_ctx* ctx = m_context ? m_context : cwl::require_context();
    cwl::check(ctx, fh_denoise(ctx, m_width, m_height, reinterpret_cast<const float*>(m_d_beauty), reinterpret_cast<const float*>(m_d_normal), reinterpret_cast<const float*>(m_d_albedo),
                               reinterpret_cast<float*>(m_d_denoised), m_upscale ? 1 : 0),
               "fh_denoise");
  }
  void wait_for_completion() const { CUDA_SYNC_CHECK(); }

 private:
  fh_ctx* m_context;
  uint32_t m_width, m_height;
  const float4* m_d_beauty;
  const float4* m_d_normal;
  const float4* m_d_albedo;
  float4* m_d_denoised;
  bool m_upscale;
};
}  // namespace fredholm
