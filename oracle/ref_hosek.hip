// ref_hosek.hip -- TEST INFRASTRUCTURE: a thin driver around the REFERENCE's own Hosek-Wilkie sky sources, compiled from
// where they lie under /root/reference (never copied into this repository):
//   fredholm/include/fredholm/arhosek.h  (host: coefficient cook, arhosek_rgb_skymodelstate_alloc_init, :145-322)
//   fredholm/modules/arhosek.cu          (device: ArHosekSkyModel_GetRadianceInternal, arhosek_tristim_skymodel_radiance, :103-127)
// Built by oracle/Makefile into oracle/_ref/libref_hosek.so (git-ignored, travels to the GPU box with the snapshot).
// The cook runs on the host anywhere; the radiance kernel runs the reference's __device__ functions on the MI355X, with
// ROCm's device libm in place of CUDA's (so comparisons against it carry a float tolerance, stated in the tests).
#include <hip/hip_runtime.h>

#include "arhosek.cu"

extern "C" {

// the 3 RGB channels of the cooked state: configs[ch][0..8], radiances[ch]
int ref_hosek_state(float turbidity, float albedo, float elevation, float* cfg27, float* rad3)
{
  ArHosekSkyModelState st = arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo, elevation);
  for (int ch = 0; ch < 3; ++ch) {
    for (int i = 0; i < 9; ++i) cfg27[9 * ch + i] = st.configs[ch][i];
    rad3[ch] = st.radiances[ch];
  }
  return 0;
}

__global__ void k_ref_radiance(ArHosekSkyModelState st, int n, const float* theta, const float* gamma, float* out3)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int ch = 0; ch < 3; ++ch) out3[3 * i + ch] = arhosek_tristim_skymodel_radiance(&st, theta[i], gamma[i], ch);
}

// radiance of n (theta, gamma) pairs through the reference's device functions; returns 0, or a HIP error code
int ref_hosek_radiance(float turbidity, float albedo, float elevation, int n, const float* theta, const float* gamma, float* out3)
{
  ArHosekSkyModelState st = arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo, elevation);
  float *d_t = nullptr, *d_g = nullptr, *d_o = nullptr;
  hipError_t e;
  if ((e = hipMalloc((void**)&d_t, 4ull * n)) != hipSuccess) return (int)e;
  if ((e = hipMalloc((void**)&d_g, 4ull * n)) != hipSuccess) return (int)e;
  if ((e = hipMalloc((void**)&d_o, 12ull * n)) != hipSuccess) return (int)e;
  (void)hipMemcpy(d_t, theta, 4ull * n, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_g, gamma, 4ull * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_ref_radiance, dim3((n + 255) / 256), dim3(256), 0, 0, st, n, d_t, d_g, d_o);
  e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out3, d_o, 12ull * n, hipMemcpyDeviceToHost);
  (void)hipFree(d_t); (void)hipFree(d_g); (void)hipFree(d_o);
  return (int)e;
}

}  // extern "C"
