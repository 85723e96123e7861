// valu_rate.hip -- issue rate of the VALU instructions the BVH node test is made of, on gfx950, at 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate.bin ; run on the GPU box.
// Prints cycles per wave-instruction as one wave sees it (s_memtime) and instructions per cycle per SIMD (wall clock x nominal 2.4 GHz is
// NOT used: the per-SIMD rate is waves x instructions / s_memtime cycles).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(int iters, float* out, unsigned long long* cycles)
{
  float a[16];
  float2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = make_float2(a[i], a[i] + 0.5f); }
  const float s = 1.0001f, o = 0.0001f;
  const float2 s2 = make_float2(s, s), o2 = make_float2(o, o);
  unsigned int q = threadIdx.x * 0x01010101u + 0x04030201u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o));
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(s2), "v"(o2));
    } else if (KIND == 2) {  // cvt_f32_ubyte + fma pairs (8 + 8)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float f;
        asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(q));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(f), "v"(s));
      }
    } else if (KIND == 3) {  // max3 / min3
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o));
    } else if (KIND == 4) {  // v_pk_mul_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
    } else if (KIND == 5) {  // cndmask with SGPR-pair condition + or
      unsigned int m = q;
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(m) : "v"(q));
      a[0] += __uint_as_float(m & 0xffu);
    } else if (KIND >= 6) {  // integer / compare / select instructions on 16 independent registers
      unsigned int* u = reinterpret_cast<unsigned int*>(a);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
        if (KIND == 7) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
        if (KIND == 8) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == 9) asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(a[i]) : "v"(q));
        if (KIND == 10) asm volatile("v_cmp_nle_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(s) : "s20", "s21");
        if (KIND == 11) asm volatile("v_cndmask_b32_e64 %0, %0, 0, s[20:21]" : "+v"(u[i]) : : "s20", "s21");
        if (KIND == 12) asm volatile("v_bfe_u32 %0, %0, 5, 3" : "+v"(u[i]));
        if (KIND == 13) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
        if (KIND == 14) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(u[i]) : "v"(q));
        if (KIND == 15) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(u[(i + 1) & 15]));
        if (KIND == 16) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&p[i])) : "v"(*reinterpret_cast<const double*>(&s2)), "v"(*reinterpret_cast<const double*>(&o2)));
        if (KIND == 17) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(u[(i + 1) & 15]));
        if (KIND == 18) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
        if (KIND == 19) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.0f;
  for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_cus)
{
  const int iters = 4096;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = n_cus * wps;  // 256-thread blocks: one wave per SIMD each
    float* out; unsigned long long* cyc;
    hipMalloc((void**)&out, sizeof(float) * blocks * 256);
    hipMalloc((void**)&cyc, sizeof(unsigned long long) * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, 16, out, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double instr = (double)iters * 16.0;
    printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction (one wave's view), %.3f instr/cycle/SIMD, kernel %.3f ms -> %.2f Ginstr/s/SIMD\n", name, wps, mean / instr,
           wps * instr / mean, ms, instr * wps / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
  }
}

int main()
{
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cus, prop.clockRate);
  run<0>("v_fma_f32", n_cus);
  run<1>("v_pk_fma_f32", n_cus);
  run<2>("cvt_ubyte+fma pairs", n_cus);
  run<3>("v_max3_f32", n_cus);
  run<4>("v_pk_mul_f32", n_cus);
  run<5>("v_lshl_or_b32", n_cus);
  run<6>("v_mul_lo_u32", n_cus);
  run<7>("v_mul_u32_u24", n_cus);
  run<8>("v_max_f32", n_cus);
  run<9>("v_cvt_f32_ubyte2", n_cus);
  run<10>("v_cmp_nle_f32 (sgpr)", n_cus);
  run<11>("v_cndmask (sgpr)", n_cus);
  run<12>("v_bfe_u32", n_cus);
  run<13>("v_xor_b32", n_cus);
  run<14>("v_lshlrev_b32 sdwa", n_cus);
  run<15>("v_mad_u32_u24", n_cus);
  run<16>("v_fma_f64", n_cus);
  run<17>("v_or3_b32", n_cus);
  run<18>("v_rcp_f32", n_cus);
  run<19>("v_sqrt_f32", n_cus);
  return 0;
}
