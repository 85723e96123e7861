// issue_peak.hip -- sustained VALU issue rates on gfx950 with the shader clock measured in the kernel, the rate at which the
// traversal kernels' own instruction mixes (the 8-wide node test, the watertight triangle test of fh_trace.h) can be issued when
// nothing but instruction issue limits them, and the rate at which the vector L1 serves the loads of a node visit (every lane its
// own line).  These are the ceilings bench.py prices the traversal kernels against (roofline.bound = "valu_issue", roofline.vl1d);
// profiles/r03_issue_peak.txt holds the output.  The run starts by checking the node test against the exact slab test in double
// precision on 16.7 M random (node, ray) pairs: it may flag a child the ray misses, never miss one it enters.
//
// Method (MI355X_MICROARCH.md, "DVFS give-back" item 6): every point is ONE launch of >= 50 ms after >= 2 s of back-to-back
// launches; every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop, so
//     clock            = d(memtime) / d(memrealtime) x 100 MHz         (median over waves)
//     instr/cycle/SIMD = waves per SIMD x instructions per wave / d(memtime)
//     Ginstr/s/SIMD    = waves per SIMD x instructions per wave / wall time of the launch (HIP events)
// The grid is n_CUs x W workgroups of 256 threads (one wave per SIMD each) and every workgroup asks for 160 KiB / W of LDS, so
// exactly W workgroups are resident per CU: W waves per SIMD, all resident from the first cycle to the last.
// Expected (MI355X_MICROARCH.md:54,473): a wave64 VALU instruction occupies its SIMD for 2 cycles = 0.5 instr/cycle/SIMD =
// 1.2 G/s per SIMD at 2.4 GHz, 4 cycles per instruction as one wave alone sees it; transcendentals 8.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I fredholm_amd/csrc tools/micro/issue_peak.hip -o tools/micro/issue_peak.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fh_trace.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cycles, ticks; };

enum Kind { FMA, MAX3, CVT_UBYTE, CNDMASK, CMP_ADDC, AND_OR, LSHL, MUL_U24, RCP, SQRT, FMA_F64, NODE_TEST, TRI_TEST,
            MUL, ADD, MAX2, SDWA_MUL, FMA_MIX, PAIR_FMA_MAX3, PAIR_FMA_CVT, PAIR_SDWA_FMA, TRIPLE_FMA_FMA_MAX3, PERM, FMA_CLAMP, ALIGNBIT, MUL_LO, LSHL_B64, LSHL_ADD_U64, PK_FMA, CVT_PK_FP8,
            BFE, L1_GATHER, L1_COALESCED, N_KINDS };
static const char* kNames[N_KINDS] = {"v_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte", "v_cndmask_b32 (sgpr mask)", "v_cmp_le_f32 + v_addc_co_u32", "v_and_or_b32", "v_lshlrev_b32", "v_mul_u32_u24",
                                      "v_rcp_f32", "v_sqrt_f32", "v_fma_f64", "node8_test (fh_trace.h)", "tri_test (fh_trace.h)",
                                      "v_mul_f32", "v_add_f32", "v_max_f32", "v_mul_f32_sdwa (byte select)", "v_fma_mix_f32 (f16 src0)", "v_fma_f32 + v_max3_f32 (1:1)",
                                      "v_fma_f32 + v_cvt_f32_ubyte (1:1)", "v_mul_f32_sdwa + v_fma_f32 (1:1)", "2 v_fma_f32 + v_max3_f32", "v_perm_b32", "v_fma_f32 clamp", "v_alignbit_b32", "v_mul_lo_u32",
                                      "v_lshlrev_b64", "v_lshl_add_u64", "v_pk_fma_f32 (2 fma)", "v_cvt_pk_f32_fp8 (2 values)", "v_bfe_u32",
                                      "global_load_dwordx4, 64 lanes in 64 L1-resident lines", "global_load_dwordx4, 64 lanes contiguous (L1-resident)"};

template <int KIND>
__global__ void __launch_bounds__(256) k_issue(int iters, float* sink, Stamp* stamps)
{
  extern __shared__ unsigned char pad_lds[];
  float a[16];
  unsigned int u[16];
  double d[8];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 0x01010101u + 0x04030201u * (i + 1); }
  for (int i = 0; i < 8; ++i) d[i] = a[i];
  const float s = 1.0001f, o = 0.0001f;
  const double s64 = 1.0001, o64 = 0.0001;
  // operands of the node / triangle test: live in registers, perturbed by the result so that nothing is hoisted
  fh::Ray8 r8;
  r8.o = fh::mk3(0.1f + threadIdx.x * 1e-3f, 0.2f, 0.3f);
  r8.inv = fh::mk3(1.7f, -2.3f, 0.9f + threadIdx.x * 1e-3f);
  r8.nx = (threadIdx.x & 1) != 0; r8.ny = (threadIdx.x & 2) != 0; r8.nz = (threadIdx.x & 4) != 0; r8.oct = (r8.nx ? 0u : 4u) | (r8.ny ? 0u : 2u) | (r8.nz ? 0u : 1u);
  uint4 n0 = make_uint4(__float_as_uint(0.05f) | 120u, __float_as_uint(0.1f) | 121u, __float_as_uint(0.2f) | 119u, 0x100u | 0x5au);
  uint4 n1 = make_uint4(u[0], u[1], u[2], u[3]), n2 = make_uint4(u[4], u[5], u[6], u[7]), n3 = make_uint4(u[8], u[9], u[10], u[11]);
  fh::RayPre rp = fh::ray_prepare(r8.o, fh::mk3(0.3f, 0.5f + threadIdx.x * 1e-3f, -0.8f));
  fh::f3 p0 = fh::mk3(0.3f, 0.1f, -0.4f), p1 = fh::mk3(0.5f, 0.2f + threadIdx.x * 1e-4f, -0.7f), p2 = fh::mk3(0.1f, 0.6f, -0.9f);
  unsigned int acc = 0;
  float facc = 0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o));
        if (KIND == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o));
        if (KIND == CVT_UBYTE) asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        if (KIND == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "s20", "s21");
        if (KIND == CMP_ADDC) asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(s) : "vcc");
        if (KIND == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
        if (KIND == LSHL) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(u[i]));
        if (KIND == MUL_U24) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        if (KIND == RCP) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));
        if (KIND == SQRT) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[i]));
        if (KIND == FMA_F64 && i < 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(s64), "v"(o64));
        if (KIND == MUL) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == ADD) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(o));
        if (KIND == MAX2) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (KIND == SDWA_MUL) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a[i]) : "v"(u[i]), "v"(s));
        if (KIND == FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(u[i]), "v"(s));
        if (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(u[(i + 2) & 15]));
        if (KIND == FMA_CLAMP) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(s), "v"(o));
        if (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        if (KIND == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        if (KIND == LSHL_B64 && i < 8) asm volatile("v_lshlrev_b64 %0, 6, %0" : "+v"(d[i]));
        if (KIND == LSHL_ADD_U64 && i < 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        if (KIND == PK_FMA && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
        if (KIND == CVT_PK_FP8 && i < 8) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(d[i]) : "v"(u[i]));
        if (KIND == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
        if (KIND == PAIR_FMA_MAX3) { if (i & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); }
        if (KIND == PAIR_FMA_CVT) { if (i & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); else asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(a[i]) : "v"(u[i])); }
        if (KIND == PAIR_SDWA_FMA) { if (i & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); else asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(a[i]) : "v"(u[i]), "v"(s)); }
        if (KIND == TRIPLE_FMA_FMA_MAX3 && i < 15) { if (i % 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(o)); }
      }
    }
    if (KIND == L1_GATHER || KIND == L1_COALESCED) {
      // what a node visit asks of the vector L1: every lane 16 bytes of its own 128-byte line (the four loads of a node go to one line per lane), against the
      // same loads with the lanes side by side; an 8 KB window every CU keeps in its L1
      const unsigned voff = KIND == L1_GATHER ? (threadIdx.x & 63u) * 128u : (threadIdx.x & 63u) * 16u;
      float4 q0, q1, q2, q3, q4, q5, q6, q7;
      asm volatile("global_load_dwordx4 %0, %8, %9\n\tglobal_load_dwordx4 %1, %8, %9 offset:16\n\tglobal_load_dwordx4 %2, %8, %9 offset:32\n\tglobal_load_dwordx4 %3, %8, %9 offset:48\n\t"
                   "global_load_dwordx4 %4, %8, %9 offset:64\n\tglobal_load_dwordx4 %5, %8, %9 offset:80\n\tglobal_load_dwordx4 %6, %8, %9 offset:96\n\tglobal_load_dwordx4 %7, %8, %9 offset:112\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"(voff), "s"(sink) : "memory");
      facc += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
    }
    if (KIND == NODE_TEST) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
        // every operand counts as modified (no instruction is issued for this): nothing of the test is loop invariant
        asm volatile("" : "+v"(n0.x), "+v"(n0.y), "+v"(n0.z), "+v"(n0.w), "+v"(n1.x), "+v"(n1.y), "+v"(n1.z), "+v"(n1.w));
        asm volatile("" : "+v"(n2.x), "+v"(n2.y), "+v"(n2.z), "+v"(n2.w), "+v"(n3.x), "+v"(n3.y), "+v"(n3.z), "+v"(n3.w));
        const uint32_t hm = fh::node8_test(r8, n0, n1, n2, n3, 1e9f);
        const uint32_t perm = fh::octant_permute(hm & (n0.w & 0xffu), r8.oct);
        acc += perm + (hm & ~n0.w);
        n1.x ^= hm; n2.y += perm; n3.z ^= acc;  // the next test depends on this one like a traversal step depends on the node before it
      }
    }
    if (KIND == TRI_TEST) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
        float t, bu, bv;
        asm volatile("" : "+v"(p0.x), "+v"(p0.y), "+v"(p0.z), "+v"(p1.x), "+v"(p1.y), "+v"(p1.z), "+v"(p2.x), "+v"(p2.y), "+v"(p2.z));
        const bool h = fh::tri_test(rp, p0, p1, p2, t, bu, bv);
        facc += h ? t + bu + bv : 0.25f;
        p0.x += facc * 1e-9f; p1.y -= facc * 1e-9f;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float res = facc + (float)acc;
  for (int i = 0; i < 16; ++i) res += a[i] + (float)u[i];
  for (int i = 0; i < 8; ++i) res += (float)d[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};
}

// The node test against the same boxes in double precision: it may flag a child the ray misses (it is conservative) but never the other way round.
// Random nodes and rays; "must" = the exact test enters the box with a relative margin of 1e-5 between entry and exit.
__global__ void k_verify_node(uint32_t n, uint32_t* out)  // out: missed, must, flagged
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t st = i * 747796405u + 2891336453u;
  auto rnd = [&]() { st = st * 747796405u + 2891336453u; uint32_t w = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u; return (w >> 22) ^ w; };
  auto rf = [&]() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); };
  fh::f3 o = fh::mk3(rf() * 4.0f - 2.0f, rf() * 4.0f - 2.0f, rf() * 4.0f - 2.0f), d = fh::mk3(rf() * 2.0f - 1.0f, rf() * 2.0f - 1.0f, rf() * 2.0f - 1.0f);
  if (i % 7 == 0) d.x = 0.0f;
  if (i % 11 == 0) d.y = -0.0f;
  if (i % 13 == 0) o = o * 1.0e4f;  // a ray from far outside
  const fh::RayPre rp = fh::ray_prepare(o, d);
  const fh::Ray8 r = fh::ray8_prepare(rp, d);
  const uint32_t e = 100u + rnd() % 30u;
  uint4 n0 = make_uint4((__float_as_uint(rf() * 2.0f - 1.0f) & ~0xffu) | e, (__float_as_uint(rf() * 2.0f - 1.0f) & ~0xffu) | (e + 1u), (__float_as_uint(rf() * 2.0f - 1.0f) & ~0xffu) | (e - 1u), rnd());
  uint4 n1 = make_uint4(rnd(), rnd(), rnd(), rnd()), n2 = make_uint4(rnd(), rnd(), rnd(), rnd()), n3 = make_uint4(rnd(), rnd(), rnd(), rnd());
  if (i % 5 == 0) { n1 = make_uint4(0u, 0u, 0u, 0u); n2.x = n2.y = 0u; n2.z = n2.w = ~0u; n3 = make_uint4(~0u, ~0u, ~0u, ~0u); }  // every child the whole node box
  const float tmax = (i & 1) ? 1e9f : ((i & 2) ? 3.0e38f : rf() * 3.0f);
  const uint32_t got = fh::node8_test(r, n0, n1, n2, n3, tmax);
  const uint32_t lo_w[3][2] = {{n1.x, n1.y}, {n1.z, n1.w}, {n2.x, n2.y}}, hi_w[3][2] = {{n2.z, n2.w}, {n3.x, n3.y}, {n3.z, n3.w}};
  const uint32_t ow[3] = {n0.x, n0.y, n0.z};
  const double ro[3] = {o.x, o.y, o.z}, ri[3] = {r.inv.x, r.inv.y, r.inv.z};
  uint32_t must = 0;
  for (int c = 0; c < 8; ++c) {
    double tn = 0.0, tf = (double)tmax;
    bool empty = false;
    for (int ax = 0; ax < 3; ++ax) {
      const double org = (double)__uint_as_float(ow[ax]), sc = ldexp(1.0, (int)(ow[ax] & 0xffu) - 127);
      const double lo = org + sc * (double)((lo_w[ax][c >> 2] >> (8 * (c & 3))) & 0xffu), hi = org + sc * (double)((hi_w[ax][c >> 2] >> (8 * (c & 3))) & 0xffu);
      if (lo > hi) empty = true;
      const double a = (lo - ro[ax]) * ri[ax], b = (hi - ro[ax]) * ri[ax];
      tn = fmax(tn, fmin(a, b)); tf = fmin(tf, fmax(a, b));
    }
    if (!empty && tn * (1.0 + 1e-5) + 1e-30 < tf * (1.0 - 1e-5)) must |= 1u << c;
  }
  if (must & ~got) atomicAdd(out, 1u);
  atomicAdd(out + 1, (uint32_t)__popc(must));
  atomicAdd(out + 2, (uint32_t)__popc(got));
}
static void verify_node()
{
  uint32_t* d; uint32_t h[3] = {0, 0, 0};
  CHECK(hipMalloc((void**)&d, 12)); CHECK(hipMemset(d, 0, 12));
  const uint32_t n = 1u << 24;
  hipLaunchKernelGGL(k_verify_node, dim3(n / 256), dim3(256), 0, 0, n, d);
  CHECK(hipMemcpy(h, d, 12, hipMemcpyDeviceToHost));
  printf("node8_test against the exact slab test on %u random (node, ray) pairs: %u tests missed a child the ray enters; children entered %.3f per test, flagged %.3f\n", n, h[0], (double)h[1] / n,
         (double)h[2] / n);
  CHECK(hipFree(d));
}

struct Result { double ms, clock_ghz, per_cycle_simd, g_per_s_simd, cycles_one_wave; };

template <int KIND>
Result run_point(int n_cus, int wps, double target_ms, double instr_per_iter, float* sink, Stamp* stamps)
{
  const int blocks = n_cus * wps;
  const size_t lds = (size_t)(160 * 1024) / (wps + 1) + 512;  // W workgroups fit a CU's 160 KiB with room to spare, W + 1 do not
  CHECK(hipFuncSetAttribute((const void*)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // calibrate the iteration count on a short launch, then one launch of >= target_ms
  int iters = 2000;
  float ms = 0.0f;
  for (int round = 0; round < 3; ++round) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), lds, 0, iters, sink, stamps);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (round < 2) { const double scale = target_ms / (ms > 1e-3 ? ms : 1e-3) * (round == 0 ? 0.2 : 1.05); iters = (int)std::min(2.0e9, std::max(2000.0, iters * scale)); }
  }
  std::vector<Stamp> h((size_t)blocks * 4);
  CHECK(hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (const Stamp& s : h) { clk.push_back((double)s.cycles / (double)s.ticks * 0.1); cyc.push_back((double)s.cycles); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double instr = instr_per_iter * iters;
  Result r;
  r.ms = ms;
  r.clock_ghz = clk[clk.size() / 2];
  // The SIMD serves its oldest wave first, so the waves of a SIMD do not share its issue slots evenly and finish one after the other: a wave's own elapsed
  // cycles say little about the SIMD's rate (their median is about half the launch at eight waves).  The rate is what the whole launch delivered: all waves'
  // instructions over the launch's wall time, and per cycle with the clock the waves measured.
  r.g_per_s_simd = wps * instr / (ms * 1e-3) / 1e9;
  r.per_cycle_simd = r.g_per_s_simd / r.clock_ghz;
  r.cycles_one_wave = cyc[cyc.size() / 2] / instr;
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return r;
}

static double g_last_cycles = 0.0, g_last_clock = 0.0;  // of the last point run (for --json)

template <int KIND>
void run_kind(int n_cus, double target_ms, float* sink, Stamp* stamps, const std::vector<int>& occ, double instr_per_iter, const char* unit)
{
  for (int wps : occ) {
    const Result r = run_point<KIND>(n_cus, wps, target_ms, instr_per_iter, sink, stamps);
    g_last_cycles = 1.0 / r.per_cycle_simd; g_last_clock = r.clock_ghz;
    printf("%-34s waves/SIMD %d: launch %6.1f ms  clock %.3f GHz  %.4f G%s/s/SIMD = %.4f %s/cycle/SIMD = %7.2f SIMD cycles per %s", kNames[KIND], wps, r.ms, r.clock_ghz, r.g_per_s_simd, unit,
           r.per_cycle_simd, unit, 1.0 / r.per_cycle_simd, unit);
    if (wps == 1) printf("  (one wave alone: %.2f cycles per %s)", r.cycles_one_wave, unit);
    printf("\n");
    fflush(stdout);
  }
}

int main(int argc, char** argv)
{
  double target_ms = 60.0;
  bool quick = false;
  // --json <path> <sha>: only the node / triangle test points at 8 waves per SIMD, written as the JSON bench.py prices the traversal kernels with
  // (profiles/r*_issue_peak.json; <sha> = first 16 hex digits of sha256(fredholm_amd/csrc/fh_trace.h), whose node8_test / tri_test this binary was compiled from)
  const char* json_path = nullptr; const char* sha = "";
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--quick")) quick = true;
    else if (!strcmp(argv[i], "--json") && i + 2 < argc) { json_path = argv[i + 1]; sha = argv[i + 2]; i += 2; }
    else target_ms = atof(argv[i]);
  }
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, nominal clock %d kHz; every point is one launch of >= %.0f ms\n", prop.gcnArchName, n_cus, prop.clockRate, target_ms);
  float* sink; Stamp* stamps;
  CHECK(hipMalloc((void**)&sink, sizeof(float) * n_cus * 8 * 256));
  CHECK(hipMalloc((void**)&stamps, sizeof(Stamp) * n_cus * 8 * 4));
  // >= 2 s of back-to-back launches first: the clock the chip holds under this kind of load
  {
    const auto lds = (size_t)(160 * 1024) / 7 + 512;
    CHECK(hipFuncSetAttribute((const void*)k_issue<FMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float total = 0.0f;
    while (total < (quick ? 300.0f : 2000.0f)) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_issue<FMA>, dim3(n_cus * 6), dim3(256), lds, 0, 200000, sink, stamps);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
    printf("warm-up: %.0f ms of v_fma_f32 launches\n", total);
  }
  const std::vector<int> all = {1, 2, 4, 6, 8}, few = {1, 6, 8}, two = {6, 8};
  verify_node();
  if (json_path) {
    const std::vector<int> eight = {7, 8};
    run_kind<NODE_TEST>(n_cus, target_ms, sink, stamps, eight, 4.0, "test");
    const double node = g_last_cycles, node_clock = g_last_clock;
    run_kind<TRI_TEST>(n_cus, target_ms, sink, stamps, eight, 4.0, "test");
    FILE* f = fopen(json_path, "w");
    if (!f) { fprintf(stderr, "cannot write %s\n", json_path); return 1; }
    fprintf(f, "{\"node8_test_simd_cycles\": %.1f, \"tri_test_simd_cycles\": %.1f, \"waves_per_simd\": 8, \"clock_ghz\": %.3f, \"launch_ms\": %.0f, \"fh_trace_h_sha256_16\": \"%s\", "
               "\"source\": \"tools/micro/issue_peak.bin --json: node8_test / tri_test of fh_trace.h, operands in registers, one launch per point, clock from s_memtime / s_memrealtime\"}\n",
            node, g_last_cycles, node_clock, target_ms, sha);
    fclose(f);
    return 0;
  }
  run_kind<FMA>(n_cus, target_ms, sink, stamps, all, 64.0, "instr");
  run_kind<NODE_TEST>(n_cus, target_ms, sink, stamps, all, 4.0, "test");
  run_kind<TRI_TEST>(n_cus, target_ms, sink, stamps, few, 4.0, "test");
  run_kind<PAIR_FMA_MAX3>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
  run_kind<TRIPLE_FMA_FMA_MAX3>(n_cus, target_ms, sink, stamps, two, 60.0, "instr");
  run_kind<PAIR_FMA_CVT>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<PAIR_SDWA_FMA>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<SDWA_MUL>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
  run_kind<FMA_MIX>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
  run_kind<MUL>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<ADD>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<MAX2>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<FMA_CLAMP>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<ALIGNBIT>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<MUL_LO>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<LSHL_B64>(n_cus, target_ms, sink, stamps, two, 32.0, "instr");
  run_kind<LSHL_ADD_U64>(n_cus, target_ms, sink, stamps, two, 32.0, "instr");
  run_kind<PK_FMA>(n_cus, target_ms, sink, stamps, two, 32.0, "instr");
  run_kind<CVT_PK_FP8>(n_cus, target_ms, sink, stamps, two, 32.0, "instr");
  run_kind<BFE>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
  run_kind<L1_GATHER>(n_cus, target_ms, sink, stamps, few, 8.0, "load");
  run_kind<L1_COALESCED>(n_cus, target_ms, sink, stamps, few, 8.0, "load");
  if (!quick) {
    run_kind<MAX3>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<CVT_UBYTE>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<CNDMASK>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<CMP_ADDC>(n_cus, target_ms, sink, stamps, few, 128.0, "instr");
    run_kind<AND_OR>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<PERM>(n_cus, target_ms, sink, stamps, two, 64.0, "instr");
    run_kind<LSHL>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<MUL_U24>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<RCP>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<SQRT>(n_cus, target_ms, sink, stamps, few, 64.0, "instr");
    run_kind<FMA_F64>(n_cus, target_ms, sink, stamps, few, 32.0, "instr");
  }
  return 0;
}
