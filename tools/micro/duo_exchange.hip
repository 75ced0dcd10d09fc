// What does a per-tick exchange between the two waves of a 128-thread workgroup cost?  Each iteration: K dependent-chain VALU
// instructions (4 chains), then wave 0 writes 4 x float4 per lane to LDS and wave 1 writes 1 x float4, one s_barrier, wave 0 reads
// 1 x float4 and wave 1 reads 4 x float4 (double-buffered by iteration parity, so one barrier per iteration is enough).
// Compared with the same loop without the exchange.  64 / 256 / 512 workgroups (4 096 / 16 384 / 32 768 envs' worth).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/duo_exchange.hip -o tools/micro/duo_exchange && tools/micro/duo_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, bool XCH> __global__ __launch_bounds__(128) void k(float* out, int iters, float a, float b) {
  __shared__ float4 xo[2][4][64];
  __shared__ float4 xc[2][64];
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  float x0 = lane * 1e-3f, x1 = lane * 2e-3f, x2 = lane * 3e-3f, x3 = lane * 4e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < K / 4; ++r) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
    }
    if (XCH) {
      const int p = i & 1;
      if (role == 0) { xo[p][0][lane] = float4{x0, x1, x2, x3}; xo[p][1][lane] = float4{x1, x2, x3, x0}; xo[p][2][lane] = float4{x2, x3, x0, x1}; xo[p][3][lane] = float4{x3, x0, x1, x2}; }
      else xc[p][lane] = float4{x0, x1, x2, x3};
      __syncthreads();
      if (role == 0) { const float4 v = xc[p][lane]; x0 += v.x * 1e-9f; x1 += v.y * 1e-9f; x2 += v.z * 1e-9f; x3 += v.w * 1e-9f; }
      else { const float4 v0 = xo[p][0][lane], v1 = xo[p][1][lane], v2 = xo[p][2][lane], v3 = xo[p][3][lane];
        x0 += (v0.x + v1.y) * 1e-9f; x1 += (v1.x + v2.y) * 1e-9f; x2 += (v2.x + v3.y) * 1e-9f; x3 += (v3.x + v0.w) * 1e-9f; }
    }
  }
  out[blockIdx.x * 128 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int K, bool XCH> double run(int blocks, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, XCH>), dim3(blocks), dim3(128), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, XCH>), dim3(blocks), dim3(128), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;  // ns per iteration
}
int main() {
  float* d; (void)hipMalloc(&d, 2048 * 128 * sizeof(float));
  for (int blocks : {64, 256, 512, 1024}) {
    const double a0 = run<160, false>(blocks, d), a1 = run<160, true>(blocks, d), b0 = run<280, false>(blocks, d), b1 = run<280, true>(blocks, d);
    printf("{\"workgroups\": %d, \"ns_per_iter_160_valu\": %.1f, \"with_exchange\": %.1f, \"ns_per_iter_280_valu\": %.1f, \"with_exchange_280\": %.1f, \"exchange_cost_ns\": %.1f}\n", blocks, a0, a1, b0, b1, a1 - a0);
  }
  return 0;
}
