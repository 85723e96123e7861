// sqrt_exhaustive.hip -- every one of the 2^32 float bit patterns through fh::sqrt_cr (fh_vec.h) and through the compiler's correctly rounded sqrtf: the two must agree
// bit for bit (NaN results: both NaN).  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I fredholm_amd/csrc -I include
//        tools/micro/sqrt_exhaustive.hip -o tools/micro/sqrt_exhaustive.bin ; run on the GPU box.  Prints the number of mismatches (0 expected) and both timings.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "fh_vec.h"

__global__ void k_check(unsigned long long* bad, unsigned int* first, unsigned int* by_exp)
{
  const unsigned int stride = gridDim.x * blockDim.x;
  unsigned long long n = 0;
  unsigned int u = blockIdx.x * blockDim.x + threadIdx.x;
  for (unsigned int it = 0; it < (1u << 31) / (stride / 2u); ++it, u += stride) {
    const float x = __uint_as_float(u);
    const float a = fh::sqrt_cr(x), b = sqrtf(x);
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    if (!same) { atomicAdd(by_exp + ((u >> 23) & 255u), 1u); if (n == 0 && atomicAdd(first + 8, 1u) < 8u) first[atomicAdd(first + 9, 1u) & 7u] = u; n++; }
  }
  if (n) atomicAdd(bad, n);
}

template <int WHICH>
__global__ void k_time(float* out)
{
  float acc = 0.0f;
  float x = 1.0f + threadIdx.x * 0.37f + blockIdx.x;
  for (int i = 0; i < 4096; ++i) { const float s = WHICH ? fh::sqrt_cr(x) : sqrtf(x); acc += s; x += s; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
  unsigned long long* bad; unsigned int* first;
  hipMalloc((void**)&bad, 8); hipMalloc((void**)&first, 64);
  hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
  const unsigned int blocks = 1u << 14, threads = 256;  // stride 2^22: 1024 iterations cover 2^32 patterns
  unsigned int* by_exp; hipMalloc((void**)&by_exp, 1024); hipMemset(by_exp, 0, 1024);
  hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, bad, first, by_exp);
  hipDeviceSynchronize();
  unsigned long long h = 0; unsigned int f[16] = {};
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 64, hipMemcpyDeviceToHost);
  printf("mismatches over all 2^32 bit patterns: %llu\n", h);
  { unsigned int e[256]; hipMemcpy(e, by_exp, 1024, hipMemcpyDeviceToHost); for (int i = 0; i < 256; ++i) if (e[i]) printf("  biased exponent %d: %u\n", i, e[i]); }
  for (int i = 0; i < 8 && i < (int)f[9]; ++i) printf("  e.g. 0x%08x\n", f[i]);
  float* out; hipMalloc((void**)&out, 4 * 2048 * 256);
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (which) hipLaunchKernelGGL(k_time<1>, dim3(2048), dim3(256), 0, 0, out); else hipLaunchKernelGGL(k_time<0>, dim3(2048), dim3(256), 0, 0, out);
    hipEventRecord(e0, 0);
    if (which) hipLaunchKernelGGL(k_time<1>, dim3(2048), dim3(256), 0, 0, out); else hipLaunchKernelGGL(k_time<0>, dim3(2048), dim3(256), 0, 0, out);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms for 2048 x 256 x 4096 dependent square roots\n", which ? "fh::sqrt_cr" : "sqrtf (compiler, correctly rounded)", ms);
  }
  return h ? 1 : 0;
}
