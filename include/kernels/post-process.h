// kernels/post-process.h -- PostProcessParams + post_process_kernel_launch with the reference's signature
// (fredholm/kernels/include/kernels/post-process.h:4-10, :126-128), forwarding to fh_post_process.
#pragma once
#include "../cwl/util.h"
#include "../fredholm/types.h"

struct PostProcessParams {
  bool use_bloom;
  float bloom_threshold;
  float bloom_sigma;
  float ISO;
  float chromatic_aberration;
};

inline void post_process_kernel_launch(const float4* beauty_in, float4* beauty_high_luminance, float4* beauty_temp, int width, int height, const PostProcessParams& params,
                                       float4* beauty_out)
{
  fh_post_params p{params.use_bloom ? 1 : 0, params.bloom_threshold, params.bloom_sigma, params.ISO, params.chromatic_aberration};
  fh_ctx* ctx = cwl::require_context();
  cwl::check(ctx, fh_post_process(ctx, reinterpret_cast<const float*>(beauty_in), reinterpret_cast<float*>(beauty_high_luminance), reinterpret_cast<float*>(beauty_temp), width, height, &p,
                                  reinterpret_cast<float*>(beauty_out)),
             "fh_post_process");
}
