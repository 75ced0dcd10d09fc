// Issue cost of packed-f32 instruction FORMS on gfx950 (4 waves per SIMD, 4 independent chains per wave, placement as in
// valu_rate.hip): plain v_pk_fma / v_pk_mul / v_pk_add, the same with op_sel swizzles (swap, broadcast), with a neg modifier,
// with an SGPR-pair source, next to scalar v_fma / v_mul / v_add.  ns per instruction per SIMD from the per-SIMD busy span.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_variants.hip -o tools/micro/pk_variants && tools/micro/pk_variants
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Stamp { unsigned long long r0, r1; unsigned hw, xcc; };
__device__ __forceinline__ unsigned long long real_clock() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY4(INS) INS(x0) INS(x1) INS(x2) INS(x3)
#define K(NAME, ASM)                                                                                                         \
  __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters, float a, float b) {                           \
    extern __shared__ char pad[];                                                                                            \
    if (threadIdx.x == 9999) pad[0] = 1;                                                                                     \
    f2 x0 = {threadIdx.x * 1e-3f, 1.f}, x1 = {threadIdx.x * 2e-3f, 2.f}, x2 = {threadIdx.x * 3e-3f, 3.f}, x3 = {threadIdx.x * 4e-3f, 4.f};   \
    f2 ca = {a, a}, cb = {b, b};                                                                                             \
    unsigned long long sa; { unsigned lo_ = __float_as_uint(a); sa = ((unsigned long long)lo_ << 32) | lo_; }               \
    sa = __builtin_amdgcn_readfirstlane((unsigned)sa) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32); \
    const unsigned long long r0 = real_clock();                                                                              \
    for (int i = 0; i < iters; ++i) {                                                                                        \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { BODY4(ASM) }                                                          \
    }                                                                                                                        \
    const unsigned long long r1 = real_clock();                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x0.y + x1.x + x1.y + x2.x + x2.y + x3.x + x3.y;                       \
    if ((threadIdx.x & 63) == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));         \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{r0, r1, hw, xcc & 0xf}; } \
  }
#define I_PKFMA(V) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(V) : "v"(ca), "v"(cb));
#define I_PKMUL(V) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(V) : "v"(ca));
#define I_PKADD(V) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(V) : "v"(cb));
#define I_PKFMA_SWAP(V) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(V) : "v"(ca), "v"(cb));
#define I_PKMUL_BC(V) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(V) : "v"(ca));
#define I_PKADD_NEG(V) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(V) : "v"(cb));
#define I_PKMUL_SGPR(V) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(V) : "s"(sa));
#define I_FMA(V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(V.x) : "v"(a), "v"(b));
#define I_MUL(V) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(V.x) : "v"(a));
#define I_ADD(V) asm volatile("v_add_f32 %0, %0, %1" : "+v"(V.x) : "v"(b));
#define I_FMA_S(V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(V.x) : "s"(a), "v"(b));
#define I_MUL_S(V) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(V.x) : "s"(a));
#define I_FMAC(V) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(V.x) : "v"(a), "v"(b));
#define I_FMAAK(V) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(V.x) : "v"(a));
#define I_MUL_LIT(V) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(V.x));
#define I_MUL_INL(V) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(V.x));
#define I_MOV(V) asm volatile("v_mov_b32 %0, %1" : "=v"(V.y) : "v"(V.x));
#define I_PKMOV(V) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(V) : "v"(ca));
#define I_MED3(V) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(V.x) : "v"(a), "v"(b));
#define I_CNDMASK(V) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(V.x) : "v"(a));
#define I_SQRT(V) asm volatile("v_sqrt_f32 %0, %0" : "+v"(V.x));
K(k_fma_s, I_FMA_S) K(k_mul_s, I_MUL_S) K(k_fmac, I_FMAC) K(k_fmaak, I_FMAAK) K(k_mul_lit, I_MUL_LIT) K(k_mul_inl, I_MUL_INL) K(k_pkfma, I_PKFMA) K(k_pkmul, I_PKMUL) K(k_pkadd, I_PKADD) K(k_pkfma_swap, I_PKFMA_SWAP) K(k_pkmul_bc, I_PKMUL_BC) K(k_pkadd_neg, I_PKADD_NEG)
K(k_pkmul_sgpr, I_PKMUL_SGPR) K(k_fma, I_FMA) K(k_mul, I_MUL) K(k_add, I_ADD) K(k_mov, I_MOV) K(k_pkmov, I_PKMOV) K(k_med3, I_MED3) K(k_cndmask, I_CNDMASK) K(k_sqrt, I_SQRT)

template <typename Kern> void run(Kern kern, const char* label, int bpc, float* d, Stamp* dst) {
  const int iters = 4096, blocks = 256 * bpc;
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
  std::vector<double> v;
  for (auto& kv : per) { unsigned long long a = ~0ull, b = 0; for (auto& s : kv.second) { a = std::min(a, s.r0); b = std::max(b, s.r1); }
    v.push_back((double)(b - a) * 10.0 / ((double)iters * 16 * 4 * kv.second.size())); }
  std::sort(v.begin(), v.end());
  printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"simds\": %zu, \"ns_per_instr_per_simd_median\": %.3f}\n", label, bpc, per.size(), v[v.size() / 2]);
  fflush(stdout);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  Stamp* st; (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    run(k_fma, "v_fma_f32", w, d, st); run(k_mul, "v_mul_f32", w, d, st); run(k_pkfma, "v_pk_fma_f32", w, d, st);
  }
  for (int w : {1, 2, 4}) {
    run(k_fma_s, "v_fma_f32 one SGPR source", w, d, st); run(k_mul_s, "v_mul_f32 SGPR source", w, d, st); run(k_fmac, "v_fmac_f32", w, d, st);
    run(k_fmaak, "v_fmaak_f32 (literal)", w, d, st); run(k_mul_lit, "v_mul_f32 literal", w, d, st); run(k_mul_inl, "v_mul_f32 inline constant", w, d, st);
  }
  for (int w : {1, 4}) {
    run(k_add, "v_add_f32", w, d, st); run(k_mov, "v_mov_b32", w, d, st);
    run(k_med3, "v_med3_f32", w, d, st); run(k_cndmask, "v_cndmask_b32", w, d, st); run(k_sqrt, "v_sqrt_f32", w, d, st);
    run(k_pkmul, "v_pk_mul_f32", w, d, st); run(k_pkadd, "v_pk_add_f32", w, d, st);
    run(k_pkfma_swap, "v_pk_fma_f32 op_sel swap", w, d, st); run(k_pkmul_bc, "v_pk_mul_f32 op_sel broadcast", w, d, st);
    run(k_pkadd_neg, "v_pk_add_f32 neg", w, d, st); run(k_pkmul_sgpr, "v_pk_mul_f32 sgpr pair", w, d, st); run(k_pkmov, "v_pk_mov_b32", w, d, st);
  }
  return 0;
}
