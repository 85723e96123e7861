// fredholm/denoiser.h -- the reference wraps the proprietary OptiX AI denoiser (fredholm/include/fredholm/denoiser.h:14-146),
// which has no counterpart here.  The class keeps the constructor / denoise() / wait_for_completion() surface the
// applications call and copies beauty to the output unchanged, so callers link and run.
#pragma once
#include <cstdint>
#include <vector>

#include "../cwl/util.h"
#include "types.h"

namespace fredholm
{
class Denoiser
{
 public:
  Denoiser(fh_ctx* /*context*/, uint32_t width, uint32_t height, const float4* d_beauty, const float4* /*d_normal*/, const float4* /*d_albedo*/, const float4* d_denoised,
           bool /*upscale*/ = false)
      : m_width(width), m_height(height), m_in(d_beauty), m_out(const_cast<float4*>(d_denoised))
  {
  }
  void denoise()
  {
    fh_ctx* ctx = cwl::require_context();
    cwl::check(ctx, fh_copy_on_device(ctx, m_out, m_in, size_t(m_width) * m_height * sizeof(float4)), "fh_copy_on_device");
  }
  void wait_for_completion() { CUDA_SYNC_CHECK(); }

 private:
  uint32_t m_width, m_height;
  const float4* m_in;
  float4* m_out;
};
}  // namespace fredholm
