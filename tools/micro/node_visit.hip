// node_visit.hip -- what a WHOLE node visit of the streaming traversal costs, and whether fetching a node with four lanes per line pays.
//
// issue_peak.hip prices the node test alone, operands in registers.  A visit of traverse_stream (fh_trace.h) is more: pop the group from the LDS stack, pick the
// child, FOUR 16-byte loads of the lane's own 64-byte node (every lane its own line: 256 L1 look-ups per wave and visit, the L1 serves one per cycle and CU), the
// test, the octant permutation, push, and the hand-over of the candidate triangles to the wave's queue.  The streaming kernels sit at 0.72-0.90 of the L1's
// look-up rate AND at 86 % of what their instruction mix can issue, so which of the two a change relieves has to be measured.  Variants:
//   LANE   the product's fetch: lane l loads its node with 4 x global_load_dwordx4
//   QUAD   four lanes per node: in load j the four lanes of a quad read the node of quad lane j, 16 bytes each (one line per quad and load: 64 look-ups per wave and
//          visit instead of 256), then a 4 x 4 transpose of 16-byte pieces inside the quad: two butterfly stages of v_cndmask_b32 with a DPP quad_perm operand,
//          32 instructions
//   PAIR   two lanes per node: half the look-ups, one butterfly stage (16 instructions)
// The visit is the product's own code (node8_test, octant_permute, GroupStack<true>, wave_inclusive_sum + scatter) on a synthetic tree of `nodes` random nodes; the next
// node index depends on the test's result, so the loads are dependent as in a traversal.  `hot` of every 16 visits go to the first 1 MB of the array (the top of a real
// tree lives in L2), the others anywhere.  One launch of >= target ms per point, waves per SIMD as the product runs them (6, 7).
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I fredholm_amd/csrc tools/micro/node_visit.hip -o tools/micro/node_visit.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fh_trace.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cycles, ticks; };
enum { LANE = 0, QUAD = 1, PAIR = 2, TEST_ONLY = 3, QUAD_ASM = 4 };
static const char* kNames[5] = {"LANE (4 loads of the lane's own node)", "QUAD (4 lanes per line, transpose: 32 v_mov_dpp + 32 v_cndmask)", "PAIR (2 lanes per line + 2x2 DPP transpose)", "no loads (node in registers)",
                                "QUAD (4 lanes per line, transpose: 32 v_cndmask_b32_dpp)"};

// one butterfly step on a pair of 16-byte registers: lo = keep ? lo : dpp(hi), hi = keep ? dpp(lo) : hi ... spelled for both results at once:
//   out_a = (lane bit clear) ? a : dpp(b)        out_b = (lane bit set) ? b : dpp(a)
// as v_cndmask_b32 with a DPP first operand (D = vcc ? S1 : dpp(S0)): eight instructions for 32 bytes
#define BUTTERFLY(CTRLSTR)                                                                                                                                   \
  asm volatile("s_mov_b64 vcc, %16\n\t"                                                                                                                      \
               "v_cndmask_b32_dpp %0, %12, %8, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                               \
               "v_cndmask_b32_dpp %1, %13, %9, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                               \
               "v_cndmask_b32_dpp %2, %14, %10, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                              \
               "v_cndmask_b32_dpp %3, %15, %11, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                              \
               "s_mov_b64 vcc, %17\n\t"                                                                                                                      \
               "v_cndmask_b32_dpp %4, %8, %12, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                               \
               "v_cndmask_b32_dpp %5, %9, %13, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                               \
               "v_cndmask_b32_dpp %6, %10, %14, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                                                              \
               "v_cndmask_b32_dpp %7, %11, %15, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf"                                                                    \
               : "=&v"(oa.x), "=&v"(oa.y), "=&v"(oa.z), "=&v"(oa.w), "=&v"(ob.x), "=&v"(ob.y), "=&v"(ob.z), "=&v"(ob.w)                                      \
               : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "s"(clear_mask), "s"(set_mask)                              \
               : "vcc")
__device__ __forceinline__ void butterfly1(uint4 a, uint4 b, uint4& oa, uint4& ob)  // across lane bit 0
{
  const unsigned long long clear_mask = 0x5555555555555555ull, set_mask = 0xaaaaaaaaaaaaaaaaull;
  BUTTERFLY("quad_perm:[1,0,3,2]");
}
__device__ __forceinline__ void butterfly2(uint4 a, uint4 b, uint4& oa, uint4& ob)  // across lane bit 1
{
  const unsigned long long clear_mask = 0x3333333333333333ull, set_mask = 0xccccccccccccccccull;
  BUTTERFLY("quad_perm:[2,3,0,1]");
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint4 dpp4(uint4 v) { return make_uint4(dpp_quad<CTRL>(v.x), dpp_quad<CTRL>(v.y), dpp_quad<CTRL>(v.z), dpp_quad<CTRL>(v.w)); }
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

// the 64 bytes of node `ni` of this lane, four 16-byte pieces
template <int MODE>
__device__ __forceinline__ void fetch_node(const uint4* nodes, uint32_t ni, uint4& n0, uint4& n1, uint4& n2, uint4& n3)
{
  const uint32_t lane = threadIdx.x & 63u;
  if (MODE == LANE) {
    const uint4* nd = (const uint4*)((const char*)nodes + ((size_t)ni << 6));
    n0 = nd[0]; n1 = nd[1]; n2 = nd[2]; n3 = nd[3];
  } else if (MODE == QUAD) {
    const uint32_t q = lane & 3u;
    // load j: the node of quad lane j, piece q
    const uint32_t i0 = dpp_quad<0x00>(ni), i1 = dpp_quad<0x55>(ni), i2 = dpp_quad<0xaa>(ni), i3 = dpp_quad<0xff>(ni);
    const uint4 r0 = *(const uint4*)((const char*)nodes + ((size_t)i0 << 6) + (q << 4));
    const uint4 r1 = *(const uint4*)((const char*)nodes + ((size_t)i1 << 6) + (q << 4));
    const uint4 r2 = *(const uint4*)((const char*)nodes + ((size_t)i2 << 6) + (q << 4));
    const uint4 r3 = *(const uint4*)((const char*)nodes + ((size_t)i3 << 6) + (q << 4));
    // transpose: lane q wants piece p of ITS node = r[q] as held by lane p.  Stage 1 exchanges across lane bit 0 / load bit 0, stage 2 across bit 1.
    const bool odd = (q & 1u) != 0u, hi = (q & 2u) != 0u;
    const uint4 a0 = sel4(odd, dpp4<0xb1>(r1), r0), a1 = sel4(odd, r1, dpp4<0xb1>(r0));  // quad_perm:[1,0,3,2]
    const uint4 a2 = sel4(odd, dpp4<0xb1>(r3), r2), a3 = sel4(odd, r3, dpp4<0xb1>(r2));
    // now even lanes hold (own r0, partner's r0, own r2, partner's r2), odd lanes (partner's r1, own r1, partner's r3, own r3): piece index = lane bit 0 of the holder of ...
    const uint4 b0 = sel4(hi, dpp4<0x4e>(a2), a0), b2 = sel4(hi, a2, dpp4<0x4e>(a0));  // quad_perm:[2,3,0,1]
    const uint4 b1 = sel4(hi, dpp4<0x4e>(a3), a1), b3 = sel4(hi, a3, dpp4<0x4e>(a1));
    n0 = b0; n1 = b1; n2 = b2; n3 = b3;
  } else if (MODE == QUAD_ASM) {
    const uint32_t q = lane & 3u;
    const uint32_t i0 = dpp_quad<0x00>(ni), i1 = dpp_quad<0x55>(ni), i2 = dpp_quad<0xaa>(ni), i3 = dpp_quad<0xff>(ni);
    const uint4 r0 = *(const uint4*)((const char*)nodes + ((size_t)i0 << 6) + (q << 4));
    const uint4 r1 = *(const uint4*)((const char*)nodes + ((size_t)i1 << 6) + (q << 4));
    const uint4 r2 = *(const uint4*)((const char*)nodes + ((size_t)i2 << 6) + (q << 4));
    const uint4 r3 = *(const uint4*)((const char*)nodes + ((size_t)i3 << 6) + (q << 4));
    uint4 a0, a1, a2, a3;
    butterfly1(r0, r1, a0, a1);   // a0 = odd ? dpp(r1) : r0,  a1 = odd ? r1 : dpp(r0)
    butterfly1(r2, r3, a2, a3);
    butterfly2(a0, a2, n0, n2);   // n0 = hi ? dpp(a2) : a0,   n2 = hi ? a2 : dpp(a0)
    butterfly2(a1, a3, n1, n3);
  } else if (MODE == PAIR) {
    const uint32_t q = lane & 1u;
    const uint32_t i0 = dpp_quad<0xa0>(ni), i1 = dpp_quad<0xf5>(ni);  // quad_perm:[0,0,2,2] and [1,1,3,3]: the node of the pair's even / odd lane
    // load j: the node of pair lane j, pieces 2 q and 2 q + 1 (32 contiguous bytes per lane: two loads, the pair covers the line)
    const uint4 r00 = *(const uint4*)((const char*)nodes + ((size_t)i0 << 6) + (q << 5)), r01 = *(const uint4*)((const char*)nodes + ((size_t)i0 << 6) + (q << 5) + 16);
    const uint4 r10 = *(const uint4*)((const char*)nodes + ((size_t)i1 << 6) + (q << 5)), r11 = *(const uint4*)((const char*)nodes + ((size_t)i1 << 6) + (q << 5) + 16);
    const bool odd = q != 0u;
    // lane q wants pieces 0..3 of its node: its own half of load q and the partner's half of load q
    const uint4 mine0 = sel4(odd, r10, r00), mine1 = sel4(odd, r11, r01);          // pieces 2q, 2q+1 of my node
    const uint4 give0 = sel4(odd, r00, r10), give1 = sel4(odd, r01, r11);          // pieces of the partner's node that I hold
    const uint4 got0 = dpp4<0xb1>(give0), got1 = dpp4<0xb1>(give1);               // the partner's: pieces 2(1-q), 2(1-q)+1 of my node
    n0 = sel4(odd, got0, mine0); n1 = sel4(odd, got1, mine1); n2 = sel4(odd, mine0, got0); n3 = sel4(odd, mine1, got1);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_visit(const uint4* nodes, uint32_t n_nodes, uint32_t hot_nodes, uint32_t hot_of_16, int iters, uint32_t* sink, Stamp* stamps, uint32_t depth)
{
  extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];  // [entry][thread] stack columns, then the waves' candidate queues
  uint32_t* queue = (uint32_t*)((char*)lds_stack + fh::lds_stack_bytes(depth)) + (threadIdx.x >> 6) * fh::kCoopQueue;
  const uint32_t lane = threadIdx.x & 63u;
  fh::GroupStack<true> stack(lds_stack, (int)depth);
  fh::Ray8 r;
  uint32_t st = (blockIdx.x * 256u + threadIdx.x) * 747796405u + 2891336453u;
  auto rnd = [&]() { st = st * 747796405u + 2891336453u; const uint32_t w = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u; return (w >> 22) ^ w; };
  r.o = fh::mk3((float)(rnd() & 1023u) * 1e-3f, (float)(rnd() & 1023u) * 1e-3f, (float)(rnd() & 1023u) * 1e-3f);
  r.inv = fh::mk3(1.0f + (float)(rnd() & 255u) * 0.01f, -1.0f - (float)(rnd() & 255u) * 0.01f, 0.5f + (float)(rnd() & 255u) * 0.01f);
  r.nx = r.inv.x < 0.0f; r.ny = r.inv.y < 0.0f; r.nz = r.inv.z < 0.0f;
  r.oct = (r.nx ? 0u : 4u) | (r.ny ? 0u : 2u) | (r.nz ? 0u : 1u);
  uint32_t ni = rnd() & (n_nodes - 1u), acc = 0, q_head = 0, q_count = 0;
  uint2 group = make_uint2(ni, 0x80000000u);
  // a few entries on the stack so that pops have something to return
  for (uint32_t k = 0; k + 1 < depth && k < 4u; ++k) stack.push(make_uint2(rnd() & (n_nodes - 1u), 0x81000000u | (rnd() & 0xffu)));
  uint4 k0 = make_uint4(__float_as_uint(0.05f) | 120u, __float_as_uint(0.1f) | 121u, __float_as_uint(0.2f) | 119u, 0x100u | 0x5au), k1 = make_uint4(rnd(), rnd(), rnd(), rnd()),
        k2 = make_uint4(rnd(), rnd(), rnd(), rnd()), k3 = make_uint4(rnd(), rnd(), rnd(), rnd());
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    // pop / child selection, as in traverse_stream
    if ((group.y & 0xff000000u) == 0u) group = stack.pop();
    const uint32_t hits_imask = group.y;
    const uint32_t bit = 31u - (uint32_t)__clz((int)(hits_imask | 0x01000000u));
    group.y &= ~(1u << bit);
    if (group.y & 0xff000000u) stack.push(group); else if (stack.sp == 0) stack.push(make_uint2(ni ^ 0x155u, 0x83000000u | (acc & 0xffu)));
    const uint32_t slot = (bit - 24u) ^ r.oct;
    ni = group.x + (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
    // where the visit goes: mostly the hot top of the tree
    const uint32_t h = rnd();
    ni = (ni ^ (h >> 4)) & (((h & 15u) < hot_of_16 ? hot_nodes : n_nodes) - 1u);  // (powers of two)
    uint4 n0, n1, n2, n3;
    if (MODE == TEST_ONLY) { asm volatile("" : "+v"(k0.w), "+v"(k1.x), "+v"(k2.y), "+v"(k3.z)); n0 = k0; n1 = k1; n2 = k2; n3 = k3; }
    else fetch_node<MODE>(nodes, ni, n0, n1, n2, n3);
    const uint32_t hm = fh::node8_test(r, n0, n1, n2, n3, 1e9f);
    const uint32_t imask = n0.w & 0xffu;
    group = make_uint2((n0.w >> 8) & (n_nodes - 1u), (fh::octant_permute(hm & imask, r.oct) << 24) | imask);
    uint2 tg = make_uint2(8u * ni, hm & ~imask & 0x11u);  // (at most two candidates per lane and visit: a real node yields ~0.3)
    // hand-over of the candidates to the wave's queue (the scan form of fh_trace.h); the queue is consumed by dropping 64 entries (the triangle tests are not part of this measurement)
    const uint32_t n_cand = (uint32_t)__popc(tg.y);
    const uint32_t incl = fh::wave_inclusive_sum(n_cand);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total != 0u) {
      uint32_t pos = q_head + q_count + incl - n_cand;
      while (tg.y) {
        const uint32_t b = (uint32_t)__ffs((int)tg.y) - 1u;
        tg.y &= tg.y - 1u;
        queue[pos & (fh::kCoopQueue - 1u)] = ((tg.x + b) << 6) | lane;
        ++pos;
      }
      q_count += total;
      while (q_count >= 64u) { acc += queue[(q_head + lane) & (fh::kCoopQueue - 1u)]; q_head = (q_head + 64u) & (fh::kCoopQueue - 1u); q_count -= 64u; }
    }
    acc += hm;
    k1.x ^= hm; k2.y += acc;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + ni + group.y + q_count;
  if (lane == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};
}

// the three fetches return the same 64 bytes
template <int MODE>
__global__ void k_check(const uint4* nodes, uint32_t n_nodes, uint32_t* bad)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ni = (i * 2654435761u) & (n_nodes - 1u);
  uint4 a0, a1, a2, a3, b0, b1, b2, b3;
  fetch_node<LANE>(nodes, ni, a0, a1, a2, a3);
  fetch_node<MODE>(nodes, ni, b0, b1, b2, b3);
  const bool same = a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w && a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w && a2.x == b2.x && a2.y == b2.y && a2.z == b2.z &&
                    a2.w == b2.w && a3.x == b3.x && a3.y == b3.y && a3.z == b3.z && a3.w == b3.w;
  if (!same) atomicAdd(bad, 1u);
}

// fh_trace.h's wave prefix sum (v_add_u32 with DPP operands) against a shuffle scan
__global__ void k_scan_check(uint32_t* bad)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t v = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u >> 28) % 9u;  // 0 .. 8, like candidate counts
  const uint32_t got = fh::wave_inclusive_sum(v);
  uint32_t want = v;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)want, off); if (lane >= (uint32_t)off) want += o; }
  if (got != want) atomicAdd(bad, 1u);
}

template <int MODE>
void run(const uint4* nodes, uint32_t n_nodes, uint32_t hot_nodes, uint32_t hot_of_16, int n_cus, int wps, double target_ms, uint32_t* sink, Stamp* stamps)
{
  const uint32_t depth = 8;
  const int blocks = n_cus * wps;
  const size_t need = fh::lds_stack_bytes(depth) + 4 * fh::kCoopQueue * 4;
  size_t lds = (size_t)(160 * 1024) / (wps + 1) + 512;  // exactly `wps` workgroups per CU
  if (lds < need) lds = need;
  CHECK(hipFuncSetAttribute((const void*)k_visit<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int iters = 2000;
  float ms = 0.0f;
  for (int round = 0; round < 3; ++round) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_visit<MODE>, dim3(blocks), dim3(256), lds, 0, nodes, n_nodes, hot_nodes, hot_of_16, iters, sink, stamps, depth);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (round < 2) { const double scale = target_ms / (ms > 1e-3 ? ms : 1e-3) * (round == 0 ? 0.2 : 1.05); iters = (int)std::min(2.0e9, std::max(2000.0, iters * scale)); }
  }
  std::vector<Stamp> h((size_t)blocks * 4);
  CHECK(hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  std::vector<double> clk;
  for (const Stamp& s : h) clk.push_back((double)s.cycles / (double)s.ticks * 0.1);
  std::sort(clk.begin(), clk.end());
  const double clock = clk[clk.size() / 2];
  const double visits_per_s_simd = (double)wps * iters / (ms * 1e-3);  // wave-level visits per second and SIMD
  hipFuncAttributes at{};
  CHECK(hipFuncGetAttributes(&at, (const void*)k_visit<MODE>));
  printf("%-46s hot %2u/16  waves/SIMD %d: launch %6.1f ms  clock %.3f GHz  %.4f M wave-visits/s/SIMD = %7.1f SIMD cycles per wave-level visit  (%d VGPRs)\n", kNames[MODE], hot_of_16, wps, ms, clock,
         visits_per_s_simd / 1e6, clock * 1e9 / visits_per_s_simd, at.numRegs);
  fflush(stdout);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
  const double target_ms = argc > 1 ? atof(argv[1]) : 60.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  const uint32_t n_nodes = 1u << 18, hot_nodes = 1u << 14;  // 16.8 MB, the size class of the wide tree of the 1 M-triangle soup (12.9 MB); 1 MB of it hot (powers of two: the index arithmetic is a mask)
  printf("device %s, %d CUs; %u nodes (%.1f MB), hot part %u nodes (%.1f MB); every point one launch of >= %.0f ms\n", prop.gcnArchName, n_cus, n_nodes, n_nodes * 64e-6, hot_nodes, hot_nodes * 64e-6, target_ms);
  std::vector<uint4> h(4 * (size_t)n_nodes);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 747796405u + 2891336453u; const uint32_t w = ((st >> ((st >> 28) + 4u)) ^ st) * 277803737u; return (w >> 22) ^ w; };
  for (uint32_t i = 0; i < n_nodes; ++i) {
    const uint32_t e = 110u + rnd() % 12u;
    float ox = (float)(rnd() & 1023u) * 1e-3f - 0.5f, oy = (float)(rnd() & 1023u) * 1e-3f - 0.5f, oz = (float)(rnd() & 1023u) * 1e-3f - 0.5f;
    uint32_t wx, wy, wz;
    memcpy(&wx, &ox, 4); memcpy(&wy, &oy, 4); memcpy(&wz, &oz, 4);
    h[4 * i] = make_uint4((wx & ~0xffu) | e, (wy & ~0xffu) | e, (wz & ~0xffu) | e, ((rnd() & (n_nodes - 1u)) << 8) | (rnd() & 0xffu));
    // lo planes small, hi planes large: most children are entered by some rays
    h[4 * i + 1] = make_uint4(rnd() & 0x3f3f3f3fu, rnd() & 0x3f3f3f3fu, rnd() & 0x3f3f3f3fu, rnd() & 0x3f3f3f3fu);
    h[4 * i + 2] = make_uint4(rnd() & 0x3f3f3f3fu, rnd() & 0x3f3f3f3fu, rnd() | 0xc0c0c0c0u, rnd() | 0xc0c0c0c0u);
    h[4 * i + 3] = make_uint4(rnd() | 0xc0c0c0c0u, rnd() | 0xc0c0c0c0u, rnd() | 0xc0c0c0c0u, rnd() | 0xc0c0c0c0u);
  }
  uint4* nodes; uint32_t* sink; Stamp* stamps; uint32_t* bad;
  CHECK(hipMalloc((void**)&nodes, h.size() * sizeof(uint4)));
  CHECK(hipMemcpy(nodes, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
  CHECK(hipMalloc((void**)&sink, sizeof(uint32_t) * n_cus * 8 * 256));
  CHECK(hipMalloc((void**)&stamps, sizeof(Stamp) * n_cus * 8 * 4));
  CHECK(hipMalloc((void**)&bad, 12)); CHECK(hipMemset(bad, 0, 12));
  hipLaunchKernelGGL(k_check<QUAD>, dim3(4096), dim3(256), 0, 0, nodes, n_nodes, bad);
  hipLaunchKernelGGL(k_check<PAIR>, dim3(4096), dim3(256), 0, 0, nodes, n_nodes, bad + 1);
  hipLaunchKernelGGL(k_check<QUAD_ASM>, dim3(4096), dim3(256), 0, 0, nodes, n_nodes, bad + 2);
  uint32_t hb[3];
  CHECK(hipMemcpy(hb, bad, 12, hipMemcpyDeviceToHost));
  printf("fetch check on %u lanes: QUAD %u, PAIR %u, QUAD (v_cndmask_b32_dpp) %u lanes with bytes that differ from the per-lane fetch\n", 4096u * 256u, hb[0], hb[1], hb[2]);
  if (hb[0] || hb[1] || hb[2]) return 1;
  CHECK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(k_scan_check, dim3(4096), dim3(256), 0, 0, bad);
  CHECK(hipMemcpy(hb, bad, 4, hipMemcpyDeviceToHost));
  printf("wave_inclusive_sum (fh_trace.h) against a shuffle scan on %u lanes: %u differ\n", 4096u * 256u, hb[0]);
  if (hb[0]) return 1;
  {  // warm-up: the clock the chip holds under this load
    float total = 0.0f;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t lds = (size_t)(160 * 1024) / 7 + 512;
    CHECK(hipFuncSetAttribute((const void*)k_visit<LANE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    while (total < 1500.0f) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_visit<LANE>, dim3(n_cus * 6), dim3(256), lds, 0, nodes, n_nodes, hot_nodes, 13u, 20000, sink, stamps, 8u);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
  }
  for (uint32_t hot : {16u, 13u, 8u})
    for (int wps : {6, 7}) {
      run<LANE>(nodes, n_nodes, hot_nodes, hot, n_cus, wps, target_ms, sink, stamps);
      run<QUAD>(nodes, n_nodes, hot_nodes, hot, n_cus, wps, target_ms, sink, stamps);
      run<QUAD_ASM>(nodes, n_nodes, hot_nodes, hot, n_cus, wps, target_ms, sink, stamps);
      run<PAIR>(nodes, n_nodes, hot_nodes, hot, n_cus, wps, target_ms, sink, stamps);
    }
  run<TEST_ONLY>(nodes, n_nodes, hot_nodes, 16u, n_cus, 6, target_ms, sink, stamps);
  run<TEST_ONLY>(nodes, n_nodes, hot_nodes, 16u, n_cus, 7, target_ms, sink, stamps);
  return 0;
}
