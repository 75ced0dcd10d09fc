// Shorter sequences than sqrt_pos's (dql_device.hpp: v_rsq_f32 + one Goldschmidt step + one residual correction, 1 + 7 instructions): are any of them the
// CORRECTLY ROUNDED square root on gfx950 — over every positive normal float32, or at least over the rotor command's domain [1e-30, 838^2]?
// Exhaustive per variant against (float)sqrt((double)x); prints the misroundings in both ranges.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/sqrt_variants.hip -o tools/micro/sqrt_variants && tools/micro/sqrt_variants
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V> __device__ __forceinline__ float sq(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  if constexpr (V == 0) {        // sqrt_pos as shipped: 1 + 7
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
  } else if constexpr (V == 1) { // residual correction alone: 1 + 4
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
  } else if constexpr (V == 2) { // Goldschmidt step on g only, the correction with the unrefined h: 1 + 6
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
  } else if constexpr (V == 3) { // two residual corrections with the unrefined h: 1 + 6
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
  } else {                       // V == 4: hardware v_sqrt_f32 alone (what a reader might expect to be enough)
    return __builtin_amdgcn_sqrtf(x);
  }
}
template <int V> __global__ void k_check(unsigned long long* out, unsigned lo, unsigned hi, unsigned dlo, unsigned dhi) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long n = 0, nd = 0; unsigned worst = 0, largest = 0;
  for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
    const float x = __uint_as_float((unsigned)b);
    const unsigned want = __float_as_uint((float)__builtin_sqrt((double)x)), got = __float_as_uint(sq<V>(x));
    if (got != want) {
      ++n; if (b >= dlo && b <= dhi) ++nd; if ((unsigned)b > largest) largest = (unsigned)b;
      const unsigned e = got > want ? got - want : want - got; if (e > worst) worst = e;
    }
  }
  if (n) { atomicAdd(&out[0], n); atomicAdd(&out[1], nd); atomicMax(&out[2], (unsigned long long)worst); atomicMax(&out[3], (unsigned long long)largest); }
}
template <int V> void run(const char* what, int instr) {
  unsigned long long* out; (void)hipMalloc(&out, 32); (void)hipMemset(out, 0, 32);
  const unsigned lo = 0x00800000u, hi = 0x7f7fffffu;
  const float dl = 1e-30f, dh = 838.0f * 838.0f;
  hipLaunchKernelGGL(k_check<V>, dim3(256 * 32), dim3(256), 0, 0, out, lo, hi, __builtin_bit_cast(unsigned, dl), __builtin_bit_cast(unsigned, dh));
  (void)hipDeviceSynchronize();
  unsigned long long h[4]; (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  printf("{\"variant\": %d, \"what\": \"%s\", \"instructions\": %d, \"inputs\": %llu, \"not_correctly_rounded\": %llu, \"of_them_in_rotor_domain_1e-30_to_838sq\": %llu, \"worst_ulp\": %llu, \"largest_bad_input\": %.9g, \"largest_bad_bits\": \"0x%08x\"}\n",
         V, what, instr, (unsigned long long)hi - lo + 1, h[0], h[1], h[2], (double)__builtin_bit_cast(float, (unsigned)h[3]), (unsigned)h[3]);
  (void)hipFree(out);
}
int main() {
  run<0>("v_rsq + Goldschmidt step on (g, h) + residual correction (sqrt_pos)", 8);
  run<1>("v_rsq + residual correction", 5);
  run<2>("v_rsq + Goldschmidt step on g only + residual correction with the unrefined h", 7);
  run<3>("v_rsq + two residual corrections with the unrefined h", 7);
  run<4>("v_sqrt_f32 alone", 1);
  return 0;
}
