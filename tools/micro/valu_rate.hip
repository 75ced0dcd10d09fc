// VALU issue rate of dependent / independent f32 FMA streams at 1, 2, 3, 4, 8 waves per SIMD (MI355X): the roofline the fused
// step kernel's instruction stream is priced against.  Build WITHOUT SLP vectorisation, or the multi-chain kernels silently
// become v_pk_fma_f32:   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CHAINS> __global__ void k(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3f + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], a, b);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same work with the chains written one after the other in the source: does the compiler interleave them by itself?
template <int CHAINS> __global__ void k_seq(float* out, int iters, float a, float b) {
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3f + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[c] = __builtin_fmaf(x[c], a, b);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run_seq(int waves_per_simd, float* d) {
  const int iters = 4096;
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_seq<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_seq<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 16 * CHAINS;
  printf("{\"chains_sequential_in_source\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_instr_per_simd\": %.4f}\n", CHAINS, waves_per_simd, ms,
         ms * 1e6 / (instr_per_wave * waves_per_simd));
}
// packed f32: one v_pk_fma_f32 does two FMAs per lane
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CHAINS> __global__ void k_pk(float* out, int iters, float a, float b) {
  f2 x[CHAINS];
  const f2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = f2{threadIdx.x * 1e-3f + c, threadIdx.x * 2e-3f + c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_elementwise_fma(x[c], a2, b2);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c].x + x[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run_pk(int waves_per_simd, float* d) {
  const int iters = 4096;
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_pk<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_pk<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 16 * CHAINS;
  printf("{\"packed_v_pk_fma_f32_chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_instr_per_simd\": %.4f}\n", CHAINS, waves_per_simd, ms,
         ms * 1e6 / (instr_per_wave * waves_per_simd));
}
template <int CHAINS> void run(int waves_per_simd, float* d) {
  const int iters = 4096;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs: one 256-thread block = one wave per SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 1e-7f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 16 * CHAINS;
  printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_instr_per_simd\": %.4f}\n", CHAINS, waves_per_simd, ms,
         ms * 1e6 / (instr_per_wave * waves_per_simd));
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 3, 4, 8}) run<1>(w, d);
  for (int w : {1, 2, 3, 4, 8}) run<4>(w, d);
  for (int w : {1, 2, 4}) run<8>(w, d);
  for (int w : {1, 2}) run_seq<4>(w, d);
  for (int w : {1, 2}) run_seq<8>(w, d);
  for (int w : {1, 2, 4}) run_pk<1>(w, d);
  for (int w : {1, 2, 4}) run_pk<4>(w, d);
  return 0;
}
