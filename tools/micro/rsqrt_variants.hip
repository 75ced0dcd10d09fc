// Can the yaw frame's 1 / sqrt(n2) (dql_device.hpp yaw_rnorm: a polynomial start + three Newton steps, 13 instructions per physics tick) be v_rsq_f32 + a
// correction, and still be a function the CPU oracle can restate bit for bit?  Candidate definitions the oracle could compute: (float)(1.0 / sqrt((double)x))
// ("dbl") and the float32 value nearest to the exact 1 / sqrt(x) ("exact", decided here in double-double arithmetic).  Exhaustive over [2^-40, 2].
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/rsqrt_variants.hip -o tools/micro/rsqrt_variants && tools/micro/rsqrt_variants
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V> __device__ __forceinline__ float rs(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  if constexpr (V == 0) return y;                       // the hardware's value alone
  if constexpr (V == 1) {                               // one Newton step, residual in plain float32: 1 + 4
    const float t = x * y, r = __builtin_fmaf(-t, y, 1.0f);
    return __builtin_fmaf(0.5f * y, r, y);
  }
  if constexpr (V == 2) {                               // one Newton step with the residual 1 - x y^2 exact to second order: 1 + 6
    const float t = x * y, te = __builtin_fmaf(x, y, -t);
    float r = __builtin_fmaf(-t, y, 1.0f);
    r = __builtin_fmaf(-te, y, r);
    return __builtin_fmaf(0.5f * y, r, y);
  }
  // V == 3: the same + the second-order term 3 r^2 / 8: 1 + 8
  const float t = x * y, te = __builtin_fmaf(x, y, -t);
  float r = __builtin_fmaf(-t, y, 1.0f);
  r = __builtin_fmaf(-te, y, r);
  const float c = __builtin_fmaf(0.375f * r, r, 0.5f * r);
  return __builtin_fmaf(y, c, y);
}
// is q (a float) the float nearest to 1 / sqrt(x)?  q is nearest iff (q - ulp/2)^2 x < 1 < (q + ulp/2)^2 x, evaluated in double (q +- ulp/2 are exact doubles, the
// products carry 2^-53 relative error against margins of at least 2^-48 for float32 inputs: a midpoint is never hit exactly, 1 / sqrt(x) being irrational or a float)
__device__ bool is_nearest(float x, float q) {
  const double lo = 0.5 * ((double)q + (double)__uint_as_float(__float_as_uint(q) - 1u)), hi = 0.5 * ((double)q + (double)__uint_as_float(__float_as_uint(q) + 1u));
  return (lo * lo) * (double)x <= 1.0 && (hi * hi) * (double)x >= 1.0;
}
template <int V> __global__ void k_check(unsigned long long* out, unsigned lo, unsigned hi) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long nd = 0, ne = 0, dd = 0;
  for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
    const float x = __uint_as_float((unsigned)b);
    const float got = rs<V>(x), dbl = (float)(1.0 / __builtin_sqrt((double)x));
    if (__float_as_uint(got) != __float_as_uint(dbl)) ++nd;
    if (!is_nearest(x, got)) ++ne;
    if (!is_nearest(x, dbl)) ++dd;
  }
  atomicAdd(&out[0], nd); atomicAdd(&out[1], ne); atomicAdd(&out[2], dd);
}
template <int V> void run(const char* what, int instr) {
  unsigned long long* out; (void)hipMalloc(&out, 24); (void)hipMemset(out, 0, 24);
  const float a = 0x1p-40f, b = 2.0f;
  const unsigned lo = __builtin_bit_cast(unsigned, a), hi = __builtin_bit_cast(unsigned, b);
  hipLaunchKernelGGL(k_check<V>, dim3(256 * 16), dim3(256), 0, 0, out, lo, hi);
  (void)hipDeviceSynchronize();
  unsigned long long h[3]; (void)hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
  printf("{\"variant\": %d, \"what\": \"%s\", \"instructions\": %d, \"inputs\": %llu, \"range\": \"[2^-40, 2]\", \"differs_from_float_of_double_expression\": %llu, \"not_nearest_float\": %llu, "
         "\"double_expression_not_nearest\": %llu}\n", V, what, instr, (unsigned long long)hi - lo + 1, h[0], h[1], h[2]);
  (void)hipFree(out);
}
int main() {
  run<0>("v_rsq_f32 alone", 1);
  run<1>("v_rsq_f32 + Newton step, float32 residual", 5);
  run<2>("v_rsq_f32 + Newton step, residual exact to second order", 7);
  run<3>("v_rsq_f32 + second-order step on the exact residual", 9);
  return 0;
}
