// What a Philox round costs at the issue port: dependent chains of v_mad_u64_u32 (the 32 x 32 -> 64 product), v_xor_b32 and, for scale, v_fma_f32 with
// VGPR operands, WPS waves per SIMD (dynamic LDS sized so that exactly WPS 256-thread blocks fit a CU); shader-clock cycles per instruction of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/micro/mad64_rate.hip -o tools/micro/mad64_rate && tools/micro/mad64_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdint>
__device__ __forceinline__ unsigned long long shader_clock() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
template <int KIND> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* cyc, int iters, unsigned m0, unsigned m1, float fa) {
  extern __shared__ char pad[];
  if (threadIdx.x == 9999) pad[0] = 1;
  unsigned a = threadIdx.x * 2654435761u + 1u, b = threadIdx.x * 40503u + 7u, c = a ^ b, d = a + b;
  unsigned mv0, mv1; asm volatile("v_mov_b32 %0, %1" : "=v"(mv0) : "s"(m0)); asm volatile("v_mov_b32 %0, %1" : "=v"(mv1) : "s"(m1));
  float x = threadIdx.x * 1e-3f, y = x + 1.0f, z = y + 1.0f, w = z + 1.0f, fv; asm volatile("v_mov_b32 %0, %1" : "=v"(fv) : "s"(fa));
  const unsigned long long c0 = shader_clock();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (KIND == 0) {        // 4 independent chains of v_mad_u64_u32 (low word fed back)
        a = (unsigned)((uint64_t)a * mv0 >> 32) ^ (unsigned)((uint64_t)a * mv0);  // compiler: one mad + xor
        b = (unsigned)((uint64_t)b * mv1 >> 32) ^ (unsigned)((uint64_t)b * mv1);
      } else if (KIND == 1) { // 4 chains of v_xor_b32
        a ^= b + 0u; b ^= c; c ^= d; d ^= a;
      } else {                // 4 chains of v_fma_f32, VGPR operands
        x = __builtin_fmaf(x, fv, y); y = __builtin_fmaf(y, fv, z); z = __builtin_fmaf(z, fv, w); w = __builtin_fmaf(w, fv, x);
      }
    }
  }
  const unsigned long long c1 = shader_clock();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(x + y + z + w);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int KIND> void run(int wps, unsigned* d, unsigned long long* dc, const char* what, double per_iter) {
  const int iters = 2048, blocks = 256 * wps;
  const size_t lds = (160 * 1024) / wps - 1024;
  (void)hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), lds, 0, d, dc, iters, 0xD2511F53u, 0xCD9E8D57u, 1.0001f);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double n = (double)iters * 16 * per_iter;
  printf("{\"stream\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_source_operation_of_the_simd\": %.2f}\n", what, wps, h[h.size() / 2] / n / wps);
}
int main() {
  unsigned* d; unsigned long long* dc;
  (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&dc, 256 * 8 * 4 * 8);
  for (int wps : {1, 2, 4}) { run<0>(wps, d, dc, "32x32->64 product + xor of its halves (2 chains)", 2); run<1>(wps, d, dc, "v_xor_b32 (4 chains)", 4); run<2>(wps, d, dc, "v_fma_f32, VGPR operands (4 chains)", 4); }
  return 0;
}
