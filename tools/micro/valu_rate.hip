// VALU issue rate on MI355X with the placement made explicit: the roofline the fused step kernel's instruction stream is
// priced against, and the reconciliation with MI355X_MICROARCH.md's table row "v_fma_f32 (wave64): 2 cyc (SIMD-32); one wave
// alone: 4".
//
// Every block is 256 threads = 4 waves (one per SIMD of its CU) and asks for enough dynamic LDS that exactly BPC blocks fit
// a CU (160 KB / BPC), so a grid of 256 * BPC blocks puts exactly BPC waves on every one of the 1024 SIMDs — the hardware
// dispatcher has no other choice, and every wave records HW_ID / XCC_ID so that the host can CHECK it (waves per SIMD: min /
// max over all SIMDs; a run whose min != max is flagged).  Each wave also reads the shader clock (s_memtime) and the 100 MHz
// wall clock (s_memrealtime) around its loop: cycles per instruction follow without assuming a clock frequency.
// Streams: CHAINS independent dependent-chains of v_fma_f32 per wave (1 = fully dependent, 8 = ample ILP); the same with
// v_pk_fma_f32 (two FMAs per lane per instruction); "two waves, different chains": BPC = 2 with one chain each.
// Build WITHOUT SLP vectorisation (or the multi-chain kernels silently become v_pk_fma_f32):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Stamp { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

__device__ __forceinline__ unsigned long long shader_clock() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ unsigned long long real_clock() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CHAINS, bool PACKED> __global__ __launch_bounds__(256) void k(float* out, Stamp* st, int iters, float a, float b) {
  extern __shared__ char pad[];  // only its size matters: it bounds the blocks per CU
  if (threadIdx.x == 9999) pad[0] = 1;
  f2 x[CHAINS];
  const f2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = f2{threadIdx.x * 1e-3f + c, threadIdx.x * 2e-3f + c};
  const unsigned long long c0 = shader_clock(), r0 = real_clock();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (PACKED) x[c] = __builtin_elementwise_fma(x[c], a2, b2);
        else x[c].x = __builtin_fmaf(x[c].x, a, b);
      }
  }
  const unsigned long long c1 = shader_clock(), r1 = real_clock();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c].x + x[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1, hw, xcc & 0xf};
  }
}

template <int CHAINS, bool PACKED> void run(int bpc, float* d, Stamp* dst, const char* label) {
  const int iters = 8192;
  const int blocks = 256 * bpc;
  const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;  // bpc blocks fit a CU, bpc + 1 do not
  (void)hipFuncSetAttribute((const void*)k<CHAINS, PACKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, PACKED>), dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, PACKED>), dim3(blocks), dim3(256), lds, 0, d, dst, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> st((size_t)blocks * 4);
  (void)hipMemcpy(st.data(), dst, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  // per SIMD: the waves it hosted, how many of them overlapped in time, and its instruction rate over the span it was busy
  struct Iv { unsigned long long r0, r1, c0, c1; };
  std::map<unsigned long long, std::vector<Iv>> per_simd;
  double clk_c = 0, clk_r = 0;
  for (const Stamp& s : st) {
    const unsigned simd = (s.hw >> 4) & 3, cu = (s.hw >> 8) & 15, sh = (s.hw >> 12) & 1, se = (s.hw >> 13) & 7;
    per_simd[((((unsigned long long)s.xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd].push_back(Iv{s.r0, s.r1, s.c0, s.c1});
    clk_c += (double)(s.c1 - s.c0); clk_r += (double)(s.r1 - s.r0) * 10.0;
  }
  const double ghz = clk_c / clk_r;
  const double instr_per_wave = (double)iters * 16 * CHAINS;
  int lo = 1 << 30, hi = 0, ov_lo = 1 << 30, ov_hi = 0;
  std::vector<double> ns_per_instr;
  for (auto& kv : per_simd) {
    auto& v = kv.second;
    lo = std::min(lo, (int)v.size()); hi = std::max(hi, (int)v.size());
    unsigned long long a = ~0ull, b = 0;
    for (const Iv& iv : v) { a = std::min(a, iv.r0); b = std::max(b, iv.r1); }
    int best = 0;  // most waves of this SIMD alive at one instant
    for (const Iv& p : v) { int n = 0; for (const Iv& q : v) n += (q.r0 <= p.r0 && p.r0 < q.r1); best = std::max(best, n); }
    ov_lo = std::min(ov_lo, best); ov_hi = std::max(ov_hi, best);
    ns_per_instr.push_back((double)(b - a) * 10.0 / (instr_per_wave * (double)v.size()));
  }
  std::sort(ns_per_instr.begin(), ns_per_instr.end());
  const double med = ns_per_instr[ns_per_instr.size() / 2], worst = ns_per_instr.back();
  printf("{\"stream\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd_intended\": %d, \"simds_used\": %zu, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d, "
         "\"concurrent_waves_per_simd_min\": %d, \"concurrent_waves_per_simd_max\": %d, \"placement_ok\": %s, \"event_ms\": %.4f, "
         "\"ns_per_instr_per_simd_median\": %.3f, \"ns_per_instr_per_simd_worst\": %.3f, \"shader_clock_GHz\": %.3f, \"cycles_per_instr_per_simd_median\": %.3f, "
         "\"ns_per_instr_per_simd_from_event\": %.3f}\n",
         label, CHAINS, bpc, per_simd.size(), lo, hi, ov_lo, ov_hi, (lo == hi && hi == bpc && ov_lo == bpc && per_simd.size() == 1024) ? "true" : "false", ms,
         med, worst, ghz, med * ghz, ms * 1e6 / (instr_per_wave * bpc));
  fflush(stdout);
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  Stamp* st; (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  for (int w : {1, 2, 4, 8}) run<1, false>(w, d, st, "v_fma_f32");
  for (int w : {1, 2, 4, 8}) run<2, false>(w, d, st, "v_fma_f32");
  for (int w : {1, 2, 4, 8}) run<4, false>(w, d, st, "v_fma_f32");
  for (int w : {1, 2, 4}) run<8, false>(w, d, st, "v_fma_f32");
  for (int w : {1, 2, 4}) run<1, true>(w, d, st, "v_pk_fma_f32");
  for (int w : {1, 2, 4}) run<4, true>(w, d, st, "v_pk_fma_f32");
  return 0;
}
