// Issue cost of the OTHER instruction forms the fused step kernel's tick is made of (compares, selects, SALU mixed into a
// VALU stream, 64-bit integer adds, conversions), measured like pk_variants.hip: 256-thread blocks pinned at W per CU by their
// LDS request (= W waves per SIMD, checked from HW_ID), 4 independent chains per wave, one asm block of 64 instructions per
// loop trip so the compiler's hazard recogniser adds nothing between them.  ns per VALU instruction per SIMD — and, round 3, the
// in-kernel shader clock of the same interval (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md "DVFS give-back"
// item 6), so that the issue cost can be stated in CYCLES: the chip does not hold its 2.4 GHz maximum under a VALU-dense load.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_forms.hip -o tools/micro/valu_forms && tools/micro/valu_forms
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Stamp { unsigned long long r0, r1, c0, c1; unsigned hw, xcc; };
__device__ __forceinline__ unsigned long long core_clock() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ unsigned long long real_clock() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#define R4(S) S S S S
#define R16(S) R4(S) R4(S) R4(S) R4(S)
// G = one group (as a string with %0..%3 the four chains, %4/%5 two VGPR constants); VPG = VALU instructions per group
#define K(NAME, G, VPG)                                                                                                      \
  __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters, float a, float b) {                           \
    extern __shared__ char pad[];                                                                                            \
    if (threadIdx.x == 9999) pad[0] = 1;                                                                                     \
    float x0 = threadIdx.x * 1e-3f, x1 = threadIdx.x * 2e-3f, x2 = threadIdx.x * 3e-3f, x3 = threadIdx.x * 4e-3f;           \
    const unsigned long long r0 = real_clock(), c0 = core_clock();                                                           \
    for (int i = 0; i < iters; ++i)                                                                                          \
      asm volatile(R16(G) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207"); \
    const unsigned long long c1 = core_clock(), r1 = real_clock();                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;                                                           \
    if ((threadIdx.x & 63) == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));         \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{r0, r1, c0, c1, hw, xcc & 0xf}; } \
  }                                                                                                                          \
  static const int NAME##_vpg = VPG;

K(k_fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n", 4)
// dependent-issue latency: ONE chain (every instruction consumes its predecessor's result), two chains, and the mixed forms of the tick
K(k_fma_dep1, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n", 4)
K(k_fma_dep2, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n", 4)
K(k_mul_dep1, "v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n", 4)
K(k_cnd_vcc, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n", 4)
K(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n", 4)
K(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n", 4)
K(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cmp_lt_f32_e64 s[20:21], %3, %4\n", 4)
// the usual pair: compare, then select on its result
K(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n", 8)
// the same with the compares hoisted two selects ahead (three SGPR pairs in flight)
K(k_cmp_cnd_far, "v_cmp_lt_f32_e64 s[20:21], %0, %4\n v_cmp_lt_f32_e64 s[22:23], %1, %4\n v_cmp_lt_f32_e64 s[24:25], %2, %4\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cndmask_b32_e64 %1, %1, %5, s[22:23]\n v_cndmask_b32_e64 %2, %2, %5, s[24:25]\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %3, %3, %4, %5\n", 8)
K(k_max, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n", 4)
K(k_med3, "v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n", 4)
K(k_and, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n", 4)
K(k_bfi, "v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5\n", 4)
K(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n", 4)
K(k_abs_mod, "v_add_f32 %0, |%0|, %4\n v_add_f32 %1, |%1|, %4\n v_add_f32 %2, |%2|, %4\n v_add_f32 %3, |%3|, %4\n", 4)   // VOP3 encoding (source modifier)
K(k_fma_salu1, "v_fma_f32 %0, %0, %4, %5\n s_mov_b32 s20, 0x1234\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s21, s20, 5\n v_fma_f32 %2, %2, %4, %5\n s_and_b64 s[22:23], s[20:21], s[24:25]\n v_fma_f32 %3, %3, %4, %5\n s_mov_b32 s24, 7\n", 4)   // one SALU per VALU
K(k_fma_salu4, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n s_and_b64 s[22:23], s[20:21], s[24:25]\n", 4)                                                 // one SALU per four VALU
K(k_fma_nop, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n s_nop 0\n", 4)
K(k_add_co, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc\n", 4)   // 64-bit integer add = carry pair
K(k_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_f32_i32 %3, %3\n", 4)
K(k_floor, "v_floor_f32 %0, %0\n v_fract_f32 %1, %1\n v_rndne_f32 %2, %2\n v_trunc_f32 %3, %3\n", 4)
K(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n", 4)
K(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n", 4)
K(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_xor_b32 %2, %2, %4\n v_or_b32 %3, %3, %4\n", 4)
K(k_readlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n", 4)

template <typename Kern> void run(Kern kern, const char* label, int vpg, int bpc, float* d, Stamp* dst) {
  const int iters = 8192, blocks = 256 * bpc;
  const size_t lds = ((size_t)(160 * 1024 / bpc) - 1024) & ~(size_t)255;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.0001f, 1e-7f);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.0001f, 1e-7f);
  (void)hipDeviceSynchronize();
  std::vector<Stamp> st((size_t)blocks * 4);
  (void)hipMemcpy(st.data(), dst, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<Stamp>> per;
  for (const Stamp& s : st) { const unsigned simd = (s.hw >> 4) & 3, cu = (s.hw >> 8) & 15, sh = (s.hw >> 12) & 1, se = (s.hw >> 13) & 7;
    per[((((unsigned long long)s.xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd].push_back(s); }
  std::vector<double> ns, ghz; int lo = 1 << 30, hi = 0;
  for (const Stamp& s : st) if (s.r1 > s.r0) ghz.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  for (auto& kv : per) { unsigned long long a = ~0ull, b = 0; for (auto& s : kv.second) { a = std::min(a, s.r0); b = std::max(b, s.r1); }
    lo = std::min(lo, (int)kv.second.size()); hi = std::max(hi, (int)kv.second.size());
    ns.push_back((double)(b - a) * 10.0 / ((double)iters * 16 * vpg * kv.second.size())); }
  std::sort(ns.begin(), ns.end());
  printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"simds\": %zu, \"placement_ok\": %s, \"ns_per_valu_instr_per_simd_median\": %.3f, \"in_kernel_clock_ghz_median\": %.3f, "
         "\"cycles_per_valu_instr_per_simd\": %.2f}\n", label, bpc, per.size(),
         (lo == bpc && hi == bpc && per.size() == 1024) ? "true" : "false", ns[ns.size() / 2], clk, ns[ns.size() / 2] * clk);
  fflush(stdout);
}
#define RUN(NAME, LABEL) run(NAME, LABEL, NAME##_vpg, w, d, st)
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  Stamp* st; (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  for (int w : {1, 2, 4}) {
    RUN(k_fma, "v_fma_f32"); RUN(k_fma_dep1, "v_fma_f32, ONE dependent chain"); RUN(k_fma_dep2, "v_fma_f32, two chains"); RUN(k_mul_dep1, "v_mul / v_add alternating, one dependent chain");
    RUN(k_cnd_vcc, "v_cndmask_b32 vcc"); RUN(k_cnd_sgpr, "v_cndmask_b32 s[pair]"); RUN(k_cmp_vcc, "v_cmp_lt_f32 -> vcc");
    RUN(k_cmp_sgpr, "v_cmp_lt_f32 -> s[pair]"); RUN(k_cmp_cnd, "v_cmp + v_cndmask back to back (per instr)"); RUN(k_cmp_cnd_far, "3 v_cmp, 3 v_cndmask, 2 v_fma (per instr)");
    RUN(k_max, "v_max_f32"); RUN(k_med3, "v_med3_f32"); RUN(k_and, "v_and_b32"); RUN(k_bfi, "v_bfi_b32"); RUN(k_mov, "v_mov_b32"); RUN(k_abs_mod, "v_add_f32 |src| (VOP3)");
    RUN(k_fma_salu1, "v_fma_f32 + 1 SALU each (per VALU)"); RUN(k_fma_salu4, "v_fma_f32 + 1 SALU per 4 (per VALU)"); RUN(k_fma_nop, "v_fma_f32 + s_nop per 4 (per VALU)");
    RUN(k_add_co, "v_add_co_u32 / v_addc_co_u32"); RUN(k_cvt, "v_cvt i32<->f32"); RUN(k_floor, "v_floor/fract/rndne/trunc"); RUN(k_rcp, "v_rcp_f32");
    RUN(k_mul_lo, "v_mul_lo_u32"); RUN(k_lshl, "v_lshl/lshr/xor/or"); RUN(k_readlane, "v_readfirstlane_b32");
  }
  return 0;
}
