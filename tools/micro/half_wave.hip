// Does a wave64 VALU instruction whose upper 32 lanes are masked off (EXEC[63:32] = 0) cost the SIMD half the issue time?  If it did, a batch that fills
// only two waves per SIMD could be flown as four half-full waves with four waves' latency hiding.  Dependent v_fma_f32 chains (CHAINS per wave), every
// lane / lanes 0-31 only, WPS waves on every SIMD (dynamic LDS sized so that exactly WPS 256-thread blocks fit a CU); shader-clock cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/micro/half_wave.hip -o tools/micro/half_wave && tools/micro/half_wave
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned long long shader_clock() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
template <int CHAINS, bool HALF> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
  extern __shared__ char pad[];
  if (threadIdx.x == 9999) pad[0] = 1;
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3f + c;
  const unsigned long long c0 = shader_clock();
  if (!HALF || (threadIdx.x & 63) < 32) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], a, b);
    }
  }
  const unsigned long long c1 = shader_clock();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int CHAINS, bool HALF> void run(int wps, float* d, unsigned long long* dc) {
  const int iters = 4096, blocks = 256 * wps;
  const size_t lds = (160 * 1024) / wps - 1024;
  hipFuncSetAttribute((const void*)k<CHAINS, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, HALF>), dim3(blocks), dim3(256), lds, 0, d, dc, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double n = (double)iters * 16 * CHAINS;
  printf("{\"chains\": %d, \"lanes\": %d, \"waves_per_simd\": %d, \"cycles_per_instr_of_a_wave_median\": %.3f, \"cycles_per_instr_of_the_simd\": %.3f}\n", CHAINS, HALF ? 32 : 64, wps,
         h[h.size() / 2] / n, h[h.size() / 2] / n / wps);
}
int main() {
  float* d; unsigned long long* dc;
  hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 256 * 8 * 4 * 8);
  for (int wps : {1, 2, 4}) { run<1, false>(wps, d, dc); run<1, true>(wps, d, dc); run<4, false>(wps, d, dc); run<4, true>(wps, d, dc); }
  return 0;
}
