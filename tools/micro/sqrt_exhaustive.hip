// Is v_rsq_f32 + one Goldschmidt step + one residual correction (the sequence LLVM lowers an IEEE float32 sqrt to when denormals are
// flushed) the CORRECTLY ROUNDED square root on gfx950 for every normal float32?  The fused step's float32 tick wants to use it for the
// four rotor commands per physics tick (1 transcendental + 7 full-rate instructions instead of v_sqrt_f32 + 8 mostly four-cycle ones),
// and the CPU oracle can only follow if the result is the one IEEE defines.  Exhaustive: all 2 130 706 432 positive normal inputs,
// against (float)sqrt((double)x) (correctly rounded: double has more than 2 x 24 + 2 bits).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/sqrt_exhaustive.hip -o tools/micro/sqrt_exhaustive && tools/micro/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float sqrt_gs(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__global__ void k_check(unsigned long long* bad, unsigned* first_bad, unsigned lo, unsigned hi, unsigned long long* per_exp) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long n = 0;
  for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
    const float x = __uint_as_float((unsigned)b);
    const float want = (float)__builtin_sqrt((double)x);
    const float got = sqrt_gs(x);
    if (__float_as_uint(got) != __float_as_uint(want)) { ++n; atomicMin(first_bad, (unsigned)b); atomicAdd(&per_exp[(unsigned)b >> 23], 1ull); }
  }
  if (n) atomicAdd(bad, n);
}
int main() {
  unsigned long long* bad; unsigned* first;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 4);
  (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0xff, 4);
  unsigned long long* pe; (void)hipMalloc(&pe, 256 * 8); (void)hipMemset(pe, 0, 256 * 8);
  const unsigned lo = 0x00800000u, hi = 0x7f7fffffu;  // smallest normal .. largest finite
  hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, bad, first, lo, hi, pe);
  (void)hipDeviceSynchronize();
  unsigned long long nb = 0; unsigned fb = 0;
  (void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&fb, first, 4, hipMemcpyDeviceToHost);
  printf("{\"inputs\": %llu, \"range_bits\": [\"0x%08x\", \"0x%08x\"], \"not_correctly_rounded\": %llu, \"first_bad_bits\": \"0x%08x\"}\n",
         (unsigned long long)hi - lo + 1, lo, hi, nb, fb);
  unsigned long long h[256]; (void)hipMemcpy(h, pe, sizeof(h), hipMemcpyDeviceToHost);
  printf("{\"bad_by_biased_exponent\": {");
  bool f = true;
  for (int e = 0; e < 256; ++e) if (h[e]) { printf("%s\"%d\": %llu", f ? "" : ", ", e, h[e]); f = false; }
  printf("}}\n");
  return 0;
}
