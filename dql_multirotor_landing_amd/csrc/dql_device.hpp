// dql_device.hpp — device-side arithmetic of the fused UAV-landing step for gfx950 (wave64).
//
// One lane owns one environment for a whole agent period: the per-env state lives in VGPRs from the
// first coalesced quad load to the last quad store; constants arrive as kernel arguments (SGPRs).
// Compiled with -ffp-contract=off: every fused multiply-add is an explicit fma_(), so the arithmetic is
// reproducible operation by operation (tests compare it bit for bit with the CPU oracle in the same dtype).
//
// Reference behaviour implemented here (paths relative to the reference repo,
// pkg = src/dql_multirotor_landing/src/dql_multirotor_landing):
//   pkg/mdp.py:149-170,257-569        discretise / check / reward / continuous_action / reset
//   pkg/double_q_learning.py:110-146  guess / predict / TD target
//   pkg/filters.py, pkg/pid.py:62-104 Kalman, Butterworth, PID
//   pkg/attitude_controller.py:94-156 SO(3) attitude law + inverse allocation
//   pkg/moving_platform.py:87-127     platform kinematics
//   pkg/observation_utils.py:99-158   relative observation + acceleration estimate
//   scripts/manager_node.py:192-368   100 Hz manager: PID inputs, command mux
//   rotors_gazebo_plugins/src/gazebo_motor_model.cpp:358-364,434-500, include/.../common.h:147-183  rotor model
//   pkg/landing_simulation_env.py:167-282  reset / step sequencing
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dql.h"

#define DQL_DEV __device__ __forceinline__
// -DDQL_MARK: section markers in the ISA listing (tools/isa_sections.py); never defined in the shipped build
#ifdef DQL_MARK
#define DQL_SECTION(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; SECTION " name ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DQL_SECTION(name) do { } while (0)
#endif
#ifdef DQL_WAVE_CLOCK  // diagnostic build (tools/exp_wave_clock.py): the wave's clock when phase DQL_WAVE_CLOCK is complete
#define DQL_MARK_T(e, k) do { if ((k) == DQL_WAVE_CLOCK) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); (e).mark = wall_clock64(); } } while (0)
#else
#define DQL_MARK_T(e, k) do { } while (0)
#endif

// -DDQL_PHASE_CLOCK: diagnostic build (tools/exp_phase_clock.py): shader cycles (s_memtime) each wave spends per phase of a launch, summed over
// the launch's periods and written to the episode log buffer instead of the masks.  Phases: 0 state load, 1 period begin (Philox, action or
// reset placement, set-point matrix), 2 physics ticks, 3 manager ticks, 4 period end (rotation, Euler pitch, discretise, check, reward, TD
// target), 5 accumulation (LDS / global atomics, ballots, wave sums), 6 state store + accumulator flush.  No waits are inserted: a memory
// latency is charged to the phase in which the wave stalls on it.
#ifdef DQL_PHASE_CLOCK
#define DQL_PHASE(e, k) do { const unsigned long long _t = __builtin_readcyclecounter(); (e).ph[k] += _t - (e).t_last; (e).t_last = _t; } while (0)
#else
#define DQL_PHASE(e, k) do { } while (0)
#endif

namespace dql {

// ---------------------------------------------------------------------------------------------
// scalar helpers (float / double)
// ---------------------------------------------------------------------------------------------
DQL_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DQL_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Correctly rounded sqrt for x = 0 or normal x (the only inputs this kernel produces): hardware v_sqrt_f32 (<= 1 ulp) plus the
// neighbour test LLVM's own expansion uses, without its denormal rescaling and inf/nan class test (9 instead of 18
// instructions; x = 0 falls through: the lower neighbour is NaN and the upper one gives fma(-tiny, 0, 0) = -0, both compares false).
DQL_DEV float sqrt_(float x) {
  float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
  const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
  y = (rm <= 0.0f) ? ym : y;
  y = (rp > 0.0f) ? yp : y;
  return y;
}
DQL_DEV double sqrt_(double a) { return __builtin_sqrt(a); }
// Correctly rounded sqrt for x >= 2^-102 in 1 transcendental + 4 full-rate instructions: v_rsq_f32 (1 ulp), g = x y, and ONE residual correction
// g + (x - g^2) (y / 2) with the residual exact in the fma.  Rounds 3 - 5 (until their last hours) ran the sequence LLVM lowers an IEEE sqrt to when denormals
// are flushed — a Goldschmidt step on (g, y / 2) before the correction, 1 + 7 instructions; the step is not needed on this hardware: EXHAUSTIVELY verified on
// gfx950, every one of the 2.13e9 positive normal float32 inputs against (float)sqrt((double)x), both sequences misround only below 2^-102 (1.80 M and 1.84 M
// inputs, largest 0x0c7fffff), where the residual goes subnormal (tools/micro/sqrt_variants.hip, profiles/r5_sqrt_variants_exhaustive.jsonl; round 3:
// tools/micro/sqrt_exhaustive.hip; dql_diag_selftest_sqrt re-runs the check on THIS function inside the library).  v_sqrt_f32 alone misrounds 15 % of all inputs.
// The tick's four rotor commands per physics tick use it on med3(w^2, 1e-30, omax^2); instead of sqrt_'s v_sqrt_f32 + 8 mostly four-cycle ones.
constexpr float SQRT_POS_MIN = 1e-30f;  // > 2^-102 = 1.97e-31
DQL_DEV float sqrt_pos(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
#ifdef DQL_AB_SQRT_GOLDSCHMIDT  // A/B builds (tools/ab_build.sh): the 1 + 7 sequence
  float g = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
#else
  const float g = x * y, h = 0.5f * y;
#endif
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
DQL_DEV float abs_(float a) { return __builtin_fabsf(a); }
DQL_DEV double abs_(double a) { return __builtin_fabs(a); }
DQL_DEV float rint_(float a) { return __builtin_rintf(a); }
DQL_DEV double rint_(double a) { return __builtin_rint(a); }
template <typename T> DQL_DEV T clip(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }
// The clips of the 500 Hz loop (PID integrator and output, rotor speed limit): v_med3_f32, one instruction instead of two
// compares and two selects.  Same value as `clip` for lo <= hi and any non-NaN x, except that a signed-zero tie resolves the
// way the instruction does (max(-0, +0) = +0); the oracle restates the instruction (oracle/dql_oracle.c: med3_f32).
template <typename T> DQL_DEV T clip3(T x, T lo, T hi) { return clip(x, lo, hi); }
template <> DQL_DEV float clip3<float>(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

template <typename T> struct Quad { T a, b, c, d; };
template <> struct alignas(16) Quad<float> { float a, b, c, d; };
template <> struct alignas(32) Quad<double> { double a, b, c, d; };

// ---------------------------------------------------------------------------------------------
// deterministic elementary functions: polynomial kernels, only + - * / fma
// ---------------------------------------------------------------------------------------------
template <typename T> DQL_DEV T det_sin_k(T x) {
  const T z = x * x;
  T r = T(1.58969099521155010221e-10);
  r = fma_(r, z, T(-2.50507602534068634195e-08));
  r = fma_(r, z, T(2.75573137070700676789e-06));
  r = fma_(r, z, T(-1.98412698298579493134e-04));
  r = fma_(r, z, T(8.33333333332248946124e-03));
  r = fma_(r, z, T(-1.66666666666666324348e-01));
  return fma_(x * z, r, x);
}
template <typename T> DQL_DEV T det_cos_k(T x) {
  const T z = x * x;
  T r = T(-1.13596475577881948265e-11);
  r = fma_(r, z, T(2.08757232129817482790e-09));
  r = fma_(r, z, T(-2.75573143513906633035e-07));
  r = fma_(r, z, T(2.48015872894767294178e-05));
  r = fma_(r, z, T(-1.38888888888741095749e-03));
  r = fma_(r, z, T(4.16666666666666019037e-02));
  return fma_(z * z, r, fma_(z, T(-0.5), T(1.0)));
}
template <typename T> struct Pio2;
template <> struct Pio2<float> { static constexpr float hi = 1.5703125f, lo = 4.8382679489661923e-4f; };
template <> struct Pio2<double> { static constexpr double hi = 1.57079632673412561417e+00, lo = 6.07710050650619224932e-11; };

template <typename T> DQL_DEV void det_sincos(T x, T& s, T& c) {
  const T fn = rint_(x * T(6.36619772367581382433e-01));
  const int n = (int)fn;
  T r = fma_(-fn, Pio2<T>::hi, x);
  r = fma_(-fn, Pio2<T>::lo, r);
  const T sk = det_sin_k(r), ck = det_cos_k(r);
  const int q = n & 3;
  s = (q == 0) ? sk : (q == 1) ? ck : (q == 2) ? -sk : -ck;
  c = (q == 0) ? ck : (q == 1) ? -sk : (q == 2) ? -ck : sk;
}
template <typename T> DQL_DEV T det_atan(T x) {
  const bool neg = x < T(0.0);
  int id;
  x = abs_(x);
  T hi = T(0.0), lo = T(0.0);
  if (x < T(0.4375)) {
    id = -1;
  } else if (x < T(1.1875)) {
    if (x < T(0.6875)) { id = 0; x = (T(2.0) * x - T(1.0)) / (T(2.0) + x); hi = T(4.63647609000806093515e-01); lo = T(2.26987774529616870924e-17); }
    else { id = 1; x = (x - T(1.0)) / (x + T(1.0)); hi = T(7.85398163397448278999e-01); lo = T(3.06161699786838301793e-17); }
  } else if (x < T(2.4375)) { id = 2; x = (x - T(1.5)) / (T(1.0) + T(1.5) * x); hi = T(9.82793723247329054082e-01); lo = T(1.39033110312309984516e-17); }
  else { id = 3; x = T(-1.0) / x; hi = T(1.57079632679489655800e+00); lo = T(6.12323399573676603587e-17); }
  const T z = x * x, w = z * z;
  if constexpr (sizeof(T) == 4) {
    // float32 (round 4): the float kernel's five coefficients (fdlibm s_atanf.c) in fused Horner form, 5 instructions instead of the double
    // kernel's eleven coefficients spelled as 22 separate multiplications and additions; same argument reduction, < 1e-7 absolute
    const T s1 = z * fma_(w, fma_(w, T(6.1687607318e-02), T(1.4253635705e-01)), T(3.3333328366e-01));
    const T s2 = w * fma_(w, T(-1.0648017377e-01), T(-1.9999158382e-01));
    T r;
    if (id < 0) r = x - x * (s1 + s2);
    else r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
  }
  const T s1 = z * (T(3.33333333333329318027e-01) + w * (T(1.42857142725034663711e-01) + w * (T(9.09088713343650656196e-02) +
               w * (T(6.66107313738753120669e-02) + w * (T(4.97687799461593236017e-02) + w * T(1.62858201153657823623e-02))))));
  const T s2 = w * (T(-1.99999999998764832476e-01) + w * (T(-1.11111104054623557880e-01) + w * (T(-7.69187620504482999495e-02) +
               w * (T(-5.83357013379057348645e-02) + w * T(-3.65315727442169155270e-02)))));
  T r;
  if (id < 0) r = x - x * (s1 + s2);
  else r = hi - ((x * (s1 + s2) - lo) - x);
  return neg ? -r : r;
}
template <typename T> DQL_DEV T det_atan2(T y, T x) {
  const T pi = T(3.14159265358979311600e+00), pio2 = T(1.57079632679489655800e+00);
  if constexpr (sizeof(T) == 4) {
    // the whole wave in front of the x axis with |y / x| below det_atan's first range bound (a yaw within 23.6 degrees: every flight that is not tumbling): the
    // operations those lanes execute in the general path below — the quotient, det_atan's unreduced polynomial, the sign — in a straight line under ONE wave-uniform
    // branch instead of five lane-divergent ones (an exec-mask save / restore and a skip branch each).  Bit-identical by construction; x == 0 lanes fail the test.
    const T q = abs_(y / x);
    if (__ballot(!(x > T(0.0) && q < T(0.4375))) == 0ull) {
      const T z = q * q, w = z * z;
      const T s1 = z * fma_(w, fma_(w, T(6.1687607318e-02), T(1.4253635705e-01)), T(3.3333328366e-01));
      const T s2 = w * fma_(w, T(-1.0648017377e-01), T(-1.9999158382e-01));
      const T a = q - q * (s1 + s2);
      return y < T(0.0) ? -a : a;
    }
  }
  if (x == T(0.0)) {
    if (y == T(0.0)) return T(0.0);
    return y > T(0.0) ? pio2 : -pio2;
  }
  const T a = det_atan(abs_(y / x));
  if (x > T(0.0)) return y < T(0.0) ? -a : a;
  return y < T(0.0) ? -(pi - a) : (pi - a);
}
DQL_DEV void split_exp(float x, int& k, float& m) {
  uint32_t b = __float_as_uint(x);
  k = (int)(b >> 23) - 127;
  m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
}
DQL_DEV void split_exp(double x, int& k, double& m) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  k = (int)(b >> 52) - 1023;
  m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
}
template <typename T> DQL_DEV T det_log(T x) {  // x in (0, 1], normal
  int k; T m;
  split_exp(x, k, m);
  if (m > T(1.41421356237309514547e+00)) { m = m * T(0.5); k += 1; }
  const T f = m - T(1.0);
  const T s = f / (T(2.0) + f);
  const T z = s * s, w = z * z;
  if constexpr (sizeof(T) == 4) {  // float32 (round 4): fdlibm e_logf.c's four coefficients, fused
    const T t1 = w * fma_(w, T(0.24279078841), T(0.40000972152));
    const T t2 = z * fma_(w, T(0.28498786688), T(0.66666662693));
    const T Rr = t2 + t1, hfsq = T(0.5) * f * f, dk = (T)k;
    return dk * T(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + Rr) + dk * T(1.90821492927058770002e-10))) - f);
  }
  const T t1 = w * (T(3.999999999940941908e-01) + w * (T(2.222219843214978396e-01) + w * T(1.531383769920937332e-01)));
  const T t2 = z * (T(6.666666666666735130e-01) + w * (T(2.857142874366239149e-01) + w * (T(1.818357216161805012e-01) +
               w * T(1.479819860511658591e-01))));
  const T Rr = t2 + t1, hfsq = T(0.5) * f * f, dk = (T)k;
  return dk * T(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + Rr) + dk * T(1.90821492927058770002e-10))) - f);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------------------------
// kv: the 10 + 10 round keys (k0 + r W0, k1 + r W1) held in registers by the caller, or null.  The key is a launch constant: left to itself the compiler hoists the
// wave-uniform key schedule out of the period loop into 20 SGPRs, which the step kernel does not have — it spills them into VGPR lanes and restores one per
// round at every call (v_readlane + a hazard s_nop + an SGPR-operand v_xor: 16 + 8 + 16 issue slots per call, two calls per period); as VGPRs filled once
// per launch they cost the v_xor alone (k_step)
DQL_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4], const uint32_t* kv = nullptr) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer multiplies are quarter rate
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ (kv ? kv[r] : k0), n2 = h0 ^ c3 ^ (kv ? kv[10 + r] : k1);
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
constexpr uint32_t STREAM_ACTION = 0u, STREAM_INIT = 0xFFFFFFFFu, STREAM_NOISE0 = 16u;
template <typename T> DQL_DEV T u24(uint32_t r) { return (T)(r >> 8) * T(5.9604644775390625e-08); }
template <typename T> DQL_DEV T u24p(uint32_t r) { return (T)((r >> 8) + 1u) * T(5.9604644775390625e-08); }
template <typename T> DQL_DEV void box_muller(uint32_t ra, uint32_t rb, T& n0, T& n1) {
  const T rad = sqrt_(T(-2.0) * det_log(u24p<T>(ra)));
  T s, c;
  det_sincos(T(6.28318530717958623200e+00) * u24<T>(rb), s, c);
  n0 = rad * c; n1 = rad * s;
}

// ---------------------------------------------------------------------------------------------
// constants (kernel argument, wave-uniform -> SGPRs).  Filled on the host by make_devc() with the same
// double -> T casts the oracle uses.
// ---------------------------------------------------------------------------------------------
// MdpK: used before / after the tick loop only; lives in device memory and is read with scalar loads AFTER the loop so
// that its ~60 values never compete with the in-loop constants for SGPRs.
// CONSTANT address space pointer to it: a load through a plain global pointer below the `asm volatile("" ::: "memory")` of period_end is no
// longer provably unclobbered, and the compiler then fetches the struct with twelve VECTOR loads per lane and period (192 B per lane into 48
// VGPRs).  Constant-address-space loads of a wave-uniform address are scalar loads
// by definition (the buffer is written by hipMemcpy between launches only, never by a kernel).
#define DQL_CONST_AS __attribute__((address_space(4)))
template <typename T> struct MdpK {
  T p_max, v_max, a_max, theta_max, delta_theta, beta, sigma_a, min_alt;
  T w_p, w_v, w_theta, w_dur, w_fail, w_succ, delta_t, f_ag, timeout_steps;
  T lim_p[5], lim_v[5], lim_a[5], angles[7];
  T inv_p_max, inv_v_max, inv_a_max, inv_theta_max, dtheta_ratio;  // float32 MDP (round 4): reciprocals of the normalising constants, delta_theta / theta_max
  T tan2_mid[3];  // float32 fused step (round 5): tan^2 of the three bin boundaries of the angle grid on either side of zero (angle_bin_from_tangent)
  double gamma;
  int working, goal_logic;
  uint32_t quirks;
};
// SimK: constants of the physics tick loop (kernel argument by value).
template <typename T> struct SimK {
  T dt, g, inv_m, I[3], inv_I[3], l, h, kf, km, lkf, kmkf, aup, adn, omax, cd, crd;
  T dtm, dtg, dtI[3];  // dt / m, dt g, dt / I: the float32 tick integrates with them (one fma per component, see FAST32)
  T oup, odn, inv_mgr_dt;  // 1 - rotor alpha (up / down), 1 / (dt manager_div): float32 tick
  T nlcd, hdt, low_z;      // -(l c_d), dt / 2, mp_top + bottom: float32 tick (round 4b)
  T kR[3], kW[3], ia, ib, ic;
  T vz_kp, vz_ki, vz_lo, vz_hi, vz_wind, vz_sp;
  T yw_kp, yw_ki, yw_lo, yw_hi, yw_wind, yw_sp;
  T bw_k1, bw_k2, bw_inv;
  T bw_b2, bw_a2, bw_a3;  // float32 tick (round 4): transposed Butterworth, 2 / denom, k2 / denom, k1 / denom
  T mp_dt, mp_top, mp_hx, mp_hy, bottom;
  T noise_p, noise_v, kal_q, kal_r, mgr_dt, mp_r, mp_w;
  T kal_pss, kal_kss;  // the Kalman covariance's fixed point in T arithmetic and its gain (host: kalman_fixed_point), NaN = none
  T p_max, theta_max, delta_theta, z_init, init_sigma;  // reset placement / set-point update (before the loop)
  int div, traj, init_uniform, working, per_env_platform, two_axis;
  uint32_t quirks;
  int noisy, kal_r_zero;  // noise_p > 0 || noise_v > 0; kal_r == 0 (host: make_simk) — the wave-uniform branches on them read these, not the floats
};

// Per-tick constants held in VECTOR registers for the duration of the tick loop.  The loop wants ~50 constants on top of its
// loop state; as wave-uniform scalars they overflow the SGPR file and every use of a spilled one costs a v_readlane.  A VALU
// operand may just as well be a VGPR: one v_mov per constant before the loop (opaque to the compiler, so it stays there).
DQL_DEV float to_vgpr(float x) { float y; asm("v_mov_b32 %0, %1" : "=v"(y) : "s"(x)); return y; }
DQL_DEV uint32_t to_vgpr(uint32_t x) { uint32_t y; asm("v_mov_b32 %0, %1" : "=v"(y) : "s"(x)); return y; }
template <typename T> struct HotK {
  T dt, g, inv_m, I[3], inv_I[3], l, h, kf, lkf, kmkf, aup, adn, omax, cd, crd;
  T dtm, dtg, dtI[3], oup, odn, nlcd, hdt, low_z;
  T kR[3], kW[3], ia, ib, ic;
  T vz_kp, vz_ki, vz_lo, vz_hi, vz_wind, vz_sp;
  T yw_kp, yw_ki, yw_lo, yw_hi, yw_wind, yw_sp;
  T bw_k1, bw_k2, bw_inv, bw_b2, bw_a2, bw_a3;
  T mp_top, mp_hx, mp_hy, bottom;
};
DQL_DEV HotK<float> make_hot(const SimK<float>& s) {
  HotK<float> h;
#define DQL_HOT(f) h.f = to_vgpr(s.f)
  DQL_HOT(dt); DQL_HOT(g); DQL_HOT(inv_m); DQL_HOT(I[0]); DQL_HOT(I[1]); DQL_HOT(I[2]); DQL_HOT(inv_I[0]); DQL_HOT(inv_I[1]); DQL_HOT(inv_I[2]);
  DQL_HOT(dtm); DQL_HOT(dtg); DQL_HOT(dtI[0]); DQL_HOT(dtI[1]); DQL_HOT(dtI[2]); DQL_HOT(oup); DQL_HOT(odn); DQL_HOT(nlcd); DQL_HOT(hdt); DQL_HOT(low_z);
  DQL_HOT(l); DQL_HOT(h); DQL_HOT(kf); DQL_HOT(lkf); DQL_HOT(kmkf); DQL_HOT(aup); DQL_HOT(adn); DQL_HOT(omax); DQL_HOT(cd); DQL_HOT(crd);
  DQL_HOT(kR[0]); DQL_HOT(kR[1]); DQL_HOT(kR[2]); DQL_HOT(kW[0]); DQL_HOT(kW[1]); DQL_HOT(kW[2]); DQL_HOT(ia); DQL_HOT(ib); DQL_HOT(ic);
  DQL_HOT(vz_kp); DQL_HOT(vz_ki); DQL_HOT(vz_lo); DQL_HOT(vz_hi); DQL_HOT(vz_wind); DQL_HOT(vz_sp);
  DQL_HOT(yw_kp); DQL_HOT(yw_ki); DQL_HOT(yw_lo); DQL_HOT(yw_hi); DQL_HOT(yw_wind); DQL_HOT(yw_sp);
  DQL_HOT(bw_k1); DQL_HOT(bw_inv); DQL_HOT(bw_b2); DQL_HOT(bw_a2); DQL_HOT(bw_a3); DQL_HOT(mp_top); DQL_HOT(mp_hx); DQL_HOT(mp_hy); DQL_HOT(bottom);
#undef DQL_HOT
  h.bw_k2 = s.bw_k2;  // only steers a wave-uniform branch
  return h;
}

// ROUND 5.  The constants of the 100 Hz manager tick and of the period's begin / end (noise, Kalman, platform step, reset placement, set-point
// update) are kernel arguments: wave-uniform, so the compiler keeps them in SGPRs — of which the step kernel has none to spare (it sits at the
// 102-register cap), so it REMATERIALISES them from the kernarg segment wherever they are used: `s_load_dword sN, s[0:1], off` followed a few
// instructions later by `s_waitcnt lgkmcnt(0)`, five times per manager tick, twice per physics tick (the two PID set-points), a dozen times
// per period — ~70 scalar loads per wave and period, each parking the wave for a scalar-cache round trip.  At two waves per SIMD nobody fills
// the gap: SQ_WAIT_ANY was 18 % of the wave cycles (profiles/r5_pmc_wave_cycles_before.json).  A VALU operand may just as well be a VGPR, and at
// <= 2 waves per SIMD the kernel has ~90 of them to spare: one opaque v_mov per constant before the period loop, and the compiler has nothing
// to rematerialise.  Wave-uniform BRANCHES on these values read the host's ints (SimK::noisy, kal_r_zero) instead.
template <typename T> DQL_DEV SimK<T> period_consts_in_vgprs(SimK<T> c) { return c; }
template <> DQL_DEV SimK<float> period_consts_in_vgprs<float>(SimK<float> c) {
#define DQL_V(f) c.f = to_vgpr(c.f)
  DQL_V(noise_p); DQL_V(noise_v); DQL_V(kal_q); DQL_V(kal_r); DQL_V(mgr_dt); DQL_V(inv_mgr_dt); DQL_V(mp_dt); DQL_V(kal_pss); DQL_V(kal_kss);
  DQL_V(p_max); DQL_V(theta_max); DQL_V(delta_theta); DQL_V(z_init); DQL_V(init_sigma); DQL_V(mp_r); DQL_V(mp_w); DQL_V(vz_sp); DQL_V(yw_sp);
#undef DQL_V
  return c;
}

// The same constants as instruction LITERALS, for the reference vehicle (dql_refk.inc, generated by tools/gen_refk.py): a literal costs
// neither a register nor an SGPR operand (which halves a VALU instruction's issue rate beside other waves,
// profiles/r2_pk_variants.jsonl).  The host selects this variant only when the context's SimK is bit-identical to the table.
#include "dql_refk.inc"
struct LitK {
#define DQL_X(n, v) static constexpr float n = v;
  DQL_REFK_SCALARS(DQL_X)
#undef DQL_X
#define DQL_A(n, a, b, c) static constexpr float n[3] = {a, b, c};
  DQL_REFK_VECTORS(DQL_A)
#undef DQL_A
  float vz_sp, yw_sp;  // per-config set-points (training -0.1 m/s, simulation env -0.4): stay run-time values
};

// ROUND 5: the reference MDP's constants (pkg/mdp.py:87-147: limits, weights, angle grid, normalisers) as instruction literals too.  MdpK is ~60
// values; read as scalars at the end of every period they evict everything else from the SGPR file (and every evicted kernel argument comes back
// as an s_load + wait), read as vectors they cost twelve loads per lane.  As literals they cost nothing, and sel5 / latest_valid_level fold on
// known limits.  What a run changes stays a kernel argument (MdpRun): working level and quirks (SimK has them), goal logic, time-out steps = t_max f_ag,
// gamma.  Selected by the host together with LitK, only when make_mdpk(cfg) equals the table bit for bit (dql_hip.hip refm_matches).
template <typename T> struct MdpRun { double gamma; T timeout_steps; int goal_logic; };
struct LitM {
#define DQL_X(n, v) static constexpr float n = v;
  DQL_REFM_SCALARS(DQL_X)
#undef DQL_X
#define DQL_L(n, a, b, c, d, e) static constexpr float n[5] = {a, b, c, d, e}; \
  static DQL_DEV float at_##n(int k) { return k == 0 ? float(a) : k == 1 ? float(b) : k == 2 ? float(c) : k == 3 ? float(d) : float(e); } \
  static DQL_DEV int level_##n(int nn, float value) { /* latest_valid_level on literal limits */ \
    int res = nn - 1; \
    if (4 < nn && (value < -float(e) || value > float(e))) res = 3; \
    if (3 < nn && (value < -float(d) || value > float(d))) res = 2; \
    if (2 < nn && (value < -float(c) || value > float(c))) res = 1; \
    if (1 < nn && (value < -float(b) || value > float(b))) res = 0; \
    return res; }
  // at_<table>(k): the per-lane level lookup as a chain of selects between LITERALS.  (sel5 on the arrays — objects in constant memory once their address is
  // taken — compiled to a divergent switch that picks the ADDRESS of element k, `s_getpc` + add + addc under an exec mask per case, and a global load from it:
  // ~30 scalar instructions and a memory round trip per lookup, eight lookups per period)
  DQL_REFM_LIMITS(DQL_L)
  DQL_REFM_RATIOS(DQL_L)  // ratio_p / ratio_v[k] = lim[k + 1] / lim[k], divided on the build host (correctly rounded float32)
#undef DQL_L
#define DQL_G(n, a, b, c, d, e, f, g) static constexpr float n[7] = {a, b, c, d, e, f, g};
  DQL_REFM_GRID(DQL_G)
#undef DQL_G
#define DQL_T(n, a, b, c) static constexpr float n[3] = {a, b, c};
  DQL_REFM_TAN2(DQL_T)
#undef DQL_T
  float timeout_steps; double gamma; int working, goal_logic; uint32_t quirks;
};

enum { FL_DONE = 1, FL_CONTACT = 2, FL_ACC_INIT = 4, FL_WAS_RESET = 8, FL_OBS_CONTACT = 16 };
enum { MODE_TRAIN = 0, MODE_EVAL = 1, MODE_EXTERNAL = 2 };

constexpr int NQ_REAL = 16;  // quads of real fields per env (64 fields)
constexpr int NF_REAL = 64, NF_INT = 7;

// per-env state in registers (field order = quad layout, see dql_field_name)
template <typename T> struct Env {
#ifdef DQL_WAVE_CLOCK
  unsigned long long mark;
#endif
#ifdef DQL_PHASE_CLOCK
  unsigned long long ph[7], t_last;
#endif
  T p[3], v[3], q[4], w[3], om[4];
  T vz_i, vz_x1, vz_x2, vz_y1, vz_y2, vz_y3, vz_state;
  T yw_i, yw_x1, yw_x2, yw_y1, yw_y2, yw_y3, yw_state;
  T pitch_sp, mp_phase, mp_x, mp_u, vf_x, kal_x_x, kal_x_P, shp_p, shp_v, shp_a, cum_x, roll_sp, mp_y;
  T mp_v, vf_y, kal_y_x, kal_y_P, mp_r, mp_w;
  T shpy_p, shpy_v, shpy_a, cum_y;
  T reward, obs_px, obs_vx, obs_ax, obs_py, obs_vy, obs_ay;
  int idx_x, idx_y, step_count, cur_check, code, flags, action;
  // level and position bin of idx_x / idx_y (registers only: unpacked once per launch by load_env, carried from period to period by period_end_with — the
  // period's end used to unpack them from the packed indices it had just packed, ~30 integer instructions of the four-cycle class per period)
  int bin_k, bin_p, bin_ky, bin_py;
};

template <typename T> DQL_DEV T sel5(const T (&a)[5], int k) {
  return k == 0 ? a[0] : k == 1 ? a[1] : k == 2 ? a[2] : k == 3 ? a[3] : a[4];
}
// the MDP's per-level tables at a lane's level k: from the constants buffer, or (LitM) as literal select chains
template <typename M> DQL_DEV auto lim_p_at(const M& m, int k) { return sel5(m.lim_p, k); }
template <typename M> DQL_DEV auto lim_v_at(const M& m, int k) { return sel5(m.lim_v, k); }
template <typename M> DQL_DEV auto lim_a_at(const M& m, int k) { return sel5(m.lim_a, k); }
DQL_DEV float lim_p_at(const LitM&, int k) { return LitM::at_lim_p(k); }
DQL_DEV float lim_v_at(const LitM&, int k) { return LitM::at_lim_v(k); }
DQL_DEV float lim_a_at(const LitM&, int k) { return LitM::at_lim_a(k); }

// ---------------------------------------------------------------------------------------------
// MDP  (pkg/mdp.py)
// ---------------------------------------------------------------------------------------------
template <typename T> DQL_DEV int latest_valid_level(const T (&lim)[5], int n, T value) {  // :149-158
  int res = n - 1;
#pragma unroll
  for (int idx = 4; idx >= 1; --idx) {
    if (idx < n && (value < -lim[idx] || value > lim[idx])) res = idx - 1;
  }
  return res;
}
template <typename M, typename T> DQL_DEV int level_p_of(const M& m, int n, T v) { return latest_valid_level(m.lim_p, n, v); }
template <typename M, typename T> DQL_DEV int level_v_of(const M& m, int n, T v) { return latest_valid_level(m.lim_v, n, v); }
template <typename M, typename T> DQL_DEV int level_a_of(const M& m, int n, T v) { return latest_valid_level(m.lim_a, n, v); }
DQL_DEV int level_p_of(const LitM&, int n, float v) { return LitM::level_lim_p(n, v); }
DQL_DEV int level_v_of(const LitM&, int n, float v) { return LitM::level_lim_v(n, v); }
DQL_DEV int level_a_of(const LitM&, int n, float v) { return LitM::level_lim_a(n, v); }
template <typename T> DQL_DEV int disc3(T v, T goal, T limit) {  // :160-170
  if (-limit <= v && v < -goal) return 0;
  if (-goal <= v && v <= goal) return 1;
  if (v <= limit) return 2;
  return -1;
}
// FLOAT32 MDP, ROUND 4: the normalising divisions by the constants p_max, v_max, a_max, theta_max are multiplications by the host's reciprocals
// (a correctly rounded float32 division is ~10 instructions, seven of them per agent period); float64 keeps the reference's divisions (G1 / G2)
template <typename T> DQL_DEV T norm_by(T x, T d, T inv) {
  if constexpr (sizeof(T) == 4) return x * inv;
  else return x / d;
}
// M: MdpK<T> (run-time constants) or LitM (the reference MDP as literals, float32); BIN_GIVEN: the angle's grid bin is passed in (angle_bin_from_tangent)
struct Bins { int k, p, v; };  // level, position bin, velocity bin: what a packed state index holds besides the acceleration and angle bins
template <bool BIN_GIVEN, typename M, typename T> DQL_DEV int discretise_impl(const M& m, T rel_p, T rel_v, T rel_a, T angle, int angle_bin, Bins* bins = nullptr) {  // :257-333
  const T cp = clip(norm_by(rel_p, m.p_max, m.inv_p_max), T(-1.0), T(1.0));
  const T cv = clip(norm_by(rel_v, m.v_max, m.inv_v_max), T(-1.0), T(1.0));
  const T ca = clip(norm_by(rel_a, m.a_max, m.inv_a_max), T(-1.0), T(1.0));
  const int n = m.working + 1;
  int k = level_p_of(m, n, cp);
  const int kv = level_v_of(m, n, cv), ka = level_a_of(m, n, ca);
  k = kv < k ? kv : k;
  k = ka < k ? ka : k;
  const T lp = lim_p_at(m, k), lv = lim_v_at(m, k), la = lim_a_at(m, k);
  T pc = m.beta, vc = m.beta, ac = m.sigma_a;
  if constexpr (__is_same(M, LitM)) { if (k < m.working) { pc = LitM::at_ratio_p(k); vc = LitM::at_ratio_v(k); } }  // the same quotients, divided on the build host (tools/gen_refk.py)
  else if (k < m.working) { pc = sel5(m.lim_p, k + 1) / lp; vc = sel5(m.lim_v, k + 1) / lv; }
  if (k == m.working) ac = ac * T(m.beta);
  const int dp = disc3(cp, lp * pc, lp);
  const int dv = disc3(cv, lv * vc, lv);
  const int da = disc3(ca, la * ac, la);
  if (dp < 0 || dv < 0 || da < 0) return -1;
  int best;
  if constexpr (BIN_GIVEN) best = angle_bin;
  else {
    const T ct = clip(angle, -T(m.theta_max), T(m.theta_max));
    best = 0; T bd = abs_(T(m.angles[0]) - ct);
#pragma unroll
    for (int i = 1; i < 7; ++i) { const T d = abs_(T(m.angles[i]) - ct); if (d < bd) { bd = d; best = i; } }
  }
  if (bins) *bins = Bins{k, dp, dv};
  return (((k * 3 + dp) * 3 + dv) * 3 + da) * 7 + best;
}
template <typename M, typename T> DQL_DEV int discretise(const M& m, T rel_p, T rel_v, T rel_a, T angle) { return discretise_impl<false>(m, rel_p, rel_v, rel_a, angle, 0); }
// FLOAT32 FUSED STEP, ROUND 5.  The step needs the Euler angle only to pick the nearest of the seven grid angles -theta_max .. theta_max
// (pkg/mdp.py:145, 318-324: argmin |grid - clip(angle)|, first minimum).  The nearest grid angle changes at the six midpoints +-(j + 1/2) step, and an
// angle given as atan2(s, c) with c >= 0 lies beyond a midpoint mu exactly when tan^2 = s^2 / c^2 exceeds tan^2(mu) on that side — so the bin is three
// comparisons of s^2 with tan2_mid[j] c^2 and a sign: ~12 instructions instead of a square root, a division, an arctangent and the argmin (~110).
// Ties go where the reference's first-minimum rule sends them (towards zero on the positive side, away from it on the negative side).  c2 = c^2 (for the
// pitch: R00^2 + R10^2 = cos^2, no root needed); c_pos: c > 0 (a roll beyond 90 deg has c <= 0: the clip puts it into the outermost bin).
template <typename M> DQL_DEV int angle_bin_from_tangent(const M& m, float s, float c2, bool c_pos) {
  const float s2 = s * s;
  const float t0 = float(m.tan2_mid[0]) * c2, t1 = float(m.tan2_mid[1]) * c2, t2 = float(m.tan2_mid[2]) * c2;
  const bool neg = s < 0.0f;
  const bool b0 = neg ? s2 >= t0 : s2 > t0, b1 = neg ? s2 >= t1 : s2 > t1, b2 = neg ? s2 >= t2 : s2 > t2;
  int c = b2 ? 3 : (b1 ? 2 : (b0 ? 1 : 0));
  if (!c_pos) c = (s != 0.0f) ? 3 : 0;
  return neg ? 3 - c : 3 + c;
}
DQL_DEV int idx_level(int idx) { return idx / DQL_STATES_PER_LEVEL; }
DQL_DEV int idx_pos(int idx) { return (idx / 63) % 3; }
DQL_DEV int idx_vel(int idx) { return (idx / 21) % 3; }

template <typename T, typename K> DQL_DEV T continuous_action(const K& m, T sp, int action) {  // :543-560
  if (action == 0) { const T t = sp + m.delta_theta; return t < m.theta_max ? t : m.theta_max; }
  if (action == 1) { const T t = sp - m.delta_theta; return t > -m.theta_max ? t : -m.theta_max; }
  return sp;
}
template <typename M, typename T>
DQL_DEV int mdp_check(const M& m, int& step_count, int& cur_check, int code, int prev_idx, int cur_idx, bool contact, T rel_p_x,
                      T rel_p_y, T abs_p_z, bool two = false, int prev_idy = -1, int cur_idy = -1,
                      const Bins* cb = nullptr, int prev_k = 0, const Bins* cby = nullptr, int prev_ky = 0) {  // :335-439
  // (cb / cby + prev_k / prev_ky: the bins of the current indices and the levels of the previous ones when the caller has them unpacked; else from the indices)
  const Bins cx = cb ? *cb : Bins{idx_level(cur_idx), idx_pos(cur_idx), idx_vel(cur_idx)};
  const Bins cyb = cby ? *cby : Bins{idx_level(cur_idy), idx_pos(cur_idy), idx_vel(cur_idy)};
  const int pkx = cb ? prev_k : idx_level(prev_idx), pky = cby ? prev_ky : idx_level(prev_idy);
  // two-axis configs (beyond the reference, B16): the goal state is the joint goal of both 1-D MDPs
  const bool goal_x = prev_idx >= 0 && cx.p == 1 && cx.v == 1;
  const bool goal_y = !two || (prev_idy >= 0 && cyb.p == 1 && cyb.v == 1);
  const bool lvl_x = pkx == m.working && cx.k == m.working;
  const bool lvl_y = !two || (pky == m.working && cyb.k == m.working);
  step_count += 1;
  if (!(m.quirks & DQL_Q_STICKY_CHECK)) code = DQL_NON_TERMINAL;
  if (contact) code = DQL_TERMINAL_CONTACT;
  else if (rel_p_x < -m.p_max || rel_p_x > m.p_max) code = DQL_TERMINAL_FLYZONE_X;
  else if (rel_p_y < -m.p_max || rel_p_y > m.p_max) code = DQL_TERMINAL_FLYZONE_Y;
  else if (abs_p_z < m.min_alt) code = DQL_TERMINAL_MINIMUM_ALTITUDE;
  else if (abs_p_z > m.p_max) code = DQL_TERMINAL_FLYZONE_Z;
  else if ((T)step_count >= m.timeout_steps) code = DQL_TERMINAL_TIMEOUT;
  else if (m.goal_logic && goal_x && goal_y) {
    if (lvl_x && lvl_y) {
      cur_check += 1;
      code = ((T)cur_check >= m.f_ag) ? DQL_TERMINAL_SUCCESS : DQL_NON_TERMINAL_SUCCESS;
    } else {
      cur_check = 0;
    }
  } else if (!(m.quirks & DQL_Q_GOAL_COUNT_KEPT)) {
    cur_check = 0;
  }
  return code;
}
template <typename M, typename T>
DQL_DEV T mdp_reward(const M& m, T& shp_p, T& shp_v, T& shp_a, T& cum, int code, int cur_idx, T rel_p, T rel_v, T angle_sp, const Bins* cb = nullptr) {  // :441-541
  const T ncp = clip(norm_by(rel_p, m.p_max, m.inv_p_max), T(-1.0), T(1.0));
  const T ncv = clip(norm_by(rel_v, m.v_max, m.inv_v_max), T(-1.0), T(1.0));
  const T npitch = norm_by(angle_sp, m.theta_max, m.inv_theta_max);
  const int k = cb ? cb->k : idx_level(cur_idx);
  const T lv = lim_v_at(m, k), la = lim_a_at(m, k);
  const T prev_p = shp_p, prev_v = shp_v, prev_a = shp_a;
  shp_p = m.w_p * abs_(ncp); shp_v = m.w_v * abs_(ncv); shp_a = m.w_theta * abs_(npitch);
  const T r_p_max = abs_(m.w_p) * lv * m.delta_t;
  const T r_v_max = abs_(m.w_v) * la * m.delta_t;
  T r_theta_max;
  if constexpr (sizeof(T) == 4) r_theta_max = abs_(m.w_theta) * m.dtheta_ratio * lv;
  else r_theta_max = abs_(m.w_theta) * (m.delta_theta / m.theta_max) * lv;
  const T r_dur_max = m.w_dur * lv * m.delta_t;
  const T r_max = r_p_max + r_v_max + r_theta_max + r_dur_max;
  const T r_p = clip(shp_p - prev_p, -r_p_max, r_p_max);
  const T r_v = clip(shp_v - prev_v, -r_v_max, r_v_max);
  const T r_theta = norm_by(m.w_theta * (abs_(shp_a) - abs_(prev_a)), m.theta_max, m.inv_theta_max) * lv;
  const T r_dur = m.w_dur * lv * m.delta_t;
  T r_term;
  if (code == DQL_NON_TERMINAL_SUCCESS || code == DQL_TERMINAL_SUCCESS) r_term = m.w_succ * r_max;
  else if (code == DQL_NON_TERMINAL && !(m.quirks & DQL_Q_FAIL_TERM_EVERY_STEP)) r_term = T(0.0);
  else r_term = m.w_fail * r_max;
  const T r_t = r_p + r_v + r_theta + r_dur + r_term;
  cum += r_t;
  return r_t;
}

// ---------------------------------------------------------------------------------------------
// agent  (pkg/double_q_learning.py)
// ---------------------------------------------------------------------------------------------
DQL_DEV int argmax3(double a, double b, double c) { int k = 0; double v = a; if (b > v) { v = b; k = 1; } if (c > v) { k = 2; } return k; }
struct QRow { double a0, a1, a2, b0, b1, b2; };  // the three action values of one state in both tables
template <typename TabPtr> DQL_DEV QRow load_qrow(TabPtr qa, TabPtr qb, int idx) {
  return QRow{qa[idx * 3], qa[idx * 3 + 1], qa[idx * 3 + 2], qb[idx * 3], qb[idx * 3 + 1], qb[idx * 3 + 2]};
}
DQL_DEV int agent_predict(const QRow& r) { return argmax3((r.a0 + r.b0) / 2, (r.a1 + r.b1) / 2, (r.a2 + r.b2) / 2); }
template <typename TabPtr> DQL_DEV int agent_predict(TabPtr qa, TabPtr qb, int idx) {  // :119-124
  const double a0 = qa[idx * 3], a1 = qa[idx * 3 + 1], a2 = qa[idx * 3 + 2];
  const double b0 = qb[idx * 3], b1 = qb[idx * 3 + 1], b2 = qb[idx * 3 + 2];
  return argmax3((a0 + b0) / 2, (a1 + b1) / 2, (a2 + b2) / 2);
}

// ---------------------------------------------------------------------------------------------
// filters / PID  (pkg/filters.py, pkg/pid.py)
// ---------------------------------------------------------------------------------------------
// FLOAT32 TICK, ROUND 3.  The float64 instantiation spells the reference's expressions out operation by operation (it is what the golden
// vectors pin and what the recorded flights of tests/golden were flown with).  The float32 instantiation — the one every throughput
// figure is measured on — is bound by VALU issue, so it takes the same formulas in their shortest correctly-rounded-per-operation form:
// products folded into the additions that consume them (fma), the integration constants dt/m, dt g, dt/I multiplied out on the host,
// two Newton steps from a second-order start for the yaw frame's normalisation.  ~27 of a tick's ~360 instructions.  The oracle's float32
// build (oracle/dql_oracle.c, ORACLE_F32) spells out the same sequence, so float32 parity stays bit for bit; float32 against float64 stays
// within the north_star tolerance (tests/test_gpu_parity.py::test_f32_kernel_vs_f64_oracle_one_period).
template <typename T> struct Fast32 { static constexpr bool on = sizeof(T) == 4; };
// x * k + y with the result in a register of its OWN choosing (VOP3 v_fma_f32 / v_fmamk_f32 with a literal), k = a member of the tick constants
// picked by a tag: the constant is a literal (LitK), a VGPR (HotK) or an SGPR (SimK) depending on the layout
struct BwInv { template <typename K> static DQL_DEV constexpr auto get(const K& c) { return c.bw_inv; } };
struct BwB2 { template <typename K> static DQL_DEV constexpr auto get(const K& c) { return c.bw_b2; } };
struct LitK;
template <typename T> struct HotK;
template <typename Tag, typename K> DQL_DEV float fma3(const K& c, float x, float y) {
  float d;
  if constexpr (__is_same(K, LitK)) {
    constexpr float k = Tag::get(K{});
    asm("v_fmamk_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(__builtin_bit_cast(int, k)), "v"(y));
  } else if constexpr (__is_same(K, HotK<float>)) {
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(Tag::get(c)), "v"(y));
  } else {
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"((float)Tag::get(c)), "v"(y));
  }
  return d;
}
template <typename Tag, typename K> DQL_DEV double fma3(const K& c, double x, double y) { return fma_((double)Tag::get(c), x, y); }
template <typename T, typename K> DQL_DEV T butterworth(const K& c, T x0, T& x1, T& x2, T& y1, T& y2, T& y3) {  // filters.py:98-109
  if constexpr (Fast32<T>::on) {
    // ROUND 4: the reference's recurrence y[n] = b (x[n] + 2 x[n-1] + x[n-2]) - a2 y[n-2] - a3 y[n-3] as a TRANSPOSED direct form:
    //   y = b x + t1;  t1 = 2b x + t2;  t2 = b x - a2 y + t3;  t3 = -a3 y
    // three states (kept in the fields x1, x2, y1; y2, y3 unused) instead of five histories, and no history shifts: the rolled tick loop paid ten
    // v_mov per tick for them.  4 instructions per filter instead of 4 + 5.
    // fma3: three-address fmas.  The compiler's two-address v_fmac ties a result to its addend's register, which here rotates the three states
    // by one register per tick — and the rolled loop then pays three v_mov per filter to rotate them back.
    const T value = fma3<BwInv>(c, x0, x1);
    x1 = fma3<BwB2>(c, x0, x2);
    x2 = fma3<BwInv>(c, x0, y1);
    if (c.bw_k2 != 0) x2 = fma_(-T(c.bw_a2), value, x2);
    y1 = -(T(c.bw_a3) * value);
    return value;
  }
  T acc = x2 + T(2.0) * x1 + x0 - c.bw_k1 * y3;
  if (c.bw_k2 != 0) acc = acc - (c.bw_k2 * y2);  // -2c^2 + 2 is exactly 0 for the reference's c = 1 (pkg/filters.py:93,106)
  const T value = c.bw_inv * acc;
  x2 = x1; x1 = x0;
  y3 = y2; y2 = y1; y1 = value;
  return value;
}
template <typename T, typename K>
DQL_DEV T pid_output(const K& c, T kp, T ki, T lo, T hi, T wind, T sp, T state, T& integ, T& x1, T& x2, T& y1, T& y2, T& y3) {
  // pid.py:62-104 with Kd = 0 (launch/drone.launch:37,51; dql_create rejects Kd != 0)
  const T e0 = sp - state;
  if constexpr (Fast32<T>::on) integ = clip3(fma_(e0, T(c.dt), integ), -wind, wind);
  else integ = clip3(integ + e0 * c.dt, -wind, wind);
  const T fe = butterworth(c, e0, x1, x2, y1, y2, y3);
  if constexpr (Fast32<T>::on) return clip3(fma_(kp, fe, ki * integ), lo, hi);
  else return clip3(kp * fe + ki * integ, lo, hi);
}
// pss / kss: P's fixed point under this very update in T arithmetic, and the gain there (make_simk iterates the same three operations on the
// host).  The covariance does not depend on the data and is never reset (B15): a second of simulated time after an engine's creation every
// env sits ON the fixed point, bit for bit, and the update would recompute the same K and the same P at every 100 Hz tick — two additions, a
// correctly rounded division (ten instructions), a subtraction and a multiplication.  When every lane of the wave is there they are skipped:
// the same values by construction, so the oracle keeps the plain update.
// r_zero: R == 0, decided by the caller from the kernel argument itself (wave-uniform scalar branch): Rm may be a VGPR copy here (period_consts_in_vgprs)
template <typename T> DQL_DEV T kalman1d(T& x, T& P, T Q, T Rm, T z, T pss, T kss, bool r_zero) {  // filters.py:19-36
  if (__ballot(!(P == pss)) == 0ull) {
    x += kss * (z - x);
    return x;
  }
  P += Q;
  if (r_zero) {  // wave-uniform (launch files: no measurement noise): P / (P + 0) is exactly 1, no division to pay for
    x += (z - x);
    P *= T(0.0);
    return x;
  }
  const T K = P / (P + Rm);
  x += K * (z - x);
  P *= (T(1.0) - K);
  return x;
}
template <typename T> DQL_DEV T kalman1d(T& x, T& P, T Q, T Rm, T z, T pss = T(-1.0), T kss = T(0.0)) { return kalman1d(x, P, Q, Rm, z, pss, kss, Rm == T(0.0)); }

// ---------------------------------------------------------------------------------------------
// simulator pieces
// ---------------------------------------------------------------------------------------------
template <typename T> DQL_DEV void quat_to_R(const T (&q)[4], T (&R)[9]) {
  const T w = q[0], x = q[1], y = q[2], z = q[3];
  if constexpr (Fast32<T>::on) {  // round 4: the factor 2 applied once to x, y, z (exact), every entry one or two fused multiply-adds: 17 instructions instead of 30
    const T x2 = x + x, y2 = y + y, z2 = z + z;
    const T t = fma_(-z, z2, T(1.0));
    R[0] = fma_(-y, y2, t); R[4] = fma_(-x, x2, t); R[8] = fma_(-x, x2, fma_(-y, y2, T(1.0)));
    const T wx2 = w * x2, wy2 = w * y2, wz2 = w * z2;
    R[1] = fma_(x, y2, -wz2); R[3] = fma_(x, y2, wz2);
    R[2] = fma_(x, z2, wy2); R[6] = fma_(x, z2, -wy2);
    R[5] = fma_(y, z2, -wx2); R[7] = fma_(y, z2, wx2);
    return;
  }
  const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  R[0] = T(1.0) - T(2.0) * (yy + zz); R[1] = T(2.0) * (xy - wz); R[2] = T(2.0) * (xz + wy);
  R[3] = T(2.0) * (xy + wz); R[4] = T(1.0) - T(2.0) * (xx + zz); R[5] = T(2.0) * (yz - wx);
  R[6] = T(2.0) * (xz - wy); R[7] = T(2.0) * (yz + wx); R[8] = T(1.0) - T(2.0) * (xx + yy);
}
// cos / sin of yaw = atan2(R10, R00): (R00, R10) / sqrt(R00^2 + R10^2).  n2 = cos^2(tilt out of the horizontal) is close
// to 1 in flight, so 1/sqrt(n2) is Newton's iteration from r0 = 1.5 - 0.5 n2 (multiplies and fmas only; converged to
// rounding for tilt < ~55 deg, degrades gracefully — never NaN — for a tumbling vehicle)
template <typename T> struct YawIters;
template <> struct YawIters<float> { static constexpr int n = 4; };
template <> struct YawIters<double> { static constexpr int n = 5; };
// float32 (FAST32): second-order start 1 + d/2 + 3 d^2 / 8, d = 1 - n2 (error 5 d^3 / 16), then THREE Newton steps (round 4: two left 2e-5 at a tilt
// of 45 deg and 0.5 % at 60 deg; three are converged to rounding up to 60 deg)
template <typename T> DQL_DEV T yaw_rnorm(T n2, T c375 = T(0.375)) {  // c375: the same 0.375 from a register of the caller's (agent_period)
  const T h = T(-0.5) * n2;
  T r;
  if constexpr (Fast32<T>::on) {
    const T d = T(1.0) - n2;
    r = fma_(fma_(c375, d, T(0.5)), d, T(1.0));
#pragma unroll
    for (int k = 0; k < 3; ++k) r = r * fma_(h * r, r, T(1.5));
  } else {
    r = fma_(T(-0.5), n2, T(1.5));
#pragma unroll
    for (int k = 0; k < YawIters<T>::n; ++k) r = r * fma_(h * r, r, T(1.5));
  }
  return r;
}
template <typename T> DQL_DEV void yaw_cs(const T (&R)[9], T& c, T& s) {
  const T r = yaw_rnorm(fma_(R[0], R[0], R[3] * R[3]));
  c = R[0] * r; s = R[3] * r;
}
// the same + what the float32 attitude law builds the yaw-free attitude from: rn = 1 / cos(tilt), ct = cos(tilt) = n2 rn
template <typename T> DQL_DEV void yaw_cs(const T (&R)[9], T& c, T& s, T& ct, T& rn, T c375 = T(0.375)) {
  const T n2 = fma_(R[0], R[0], R[3] * R[3]);
  rn = yaw_rnorm(n2, c375);
  c = R[0] * rn; s = R[3] * rn; ct = n2 * rn;
}
// attitude_controller.py:107-156
// float32 (Fast32): E = R_des^T R with R_des = Rz(yaw) B and R = Rz(yaw) A, A = Ry(pitch) Rx(roll) the yaw-free attitude, is B^T A — and A needs
// no yaw at all: its last row is R's, A00 = cos(pitch) = sqrt(R00^2 + R10^2) = ct, A10 = 0, sin / cos(roll) = R21 / ct, R22 / ct = R7 rn, R8 rn,
// sin(pitch) = -R20.  5 multiplications + 19 for the seven entries instead of 2 + 12 (R_des) + 21; cy, sy are only needed by the manager tick.
// float32 rotor command: sqrt of the allocated w^2 clamped from BOTH sides in one v_med3 — below at 1e-30 (a rotor commanded to stop is commanded
// to 1e-15 rad/s: sqrt_pos's domain), above at omax^2, which replaces the motor model's min(cmd, omax) (gazebo_motor_model.cpp:358-364): for a
// correctly rounded, hence monotone, square root and an omax whose square is a float (838^2) min(sqrt(x), omax) == sqrt(min(x, omax^2)) bit for bit.
// rotor_filter<PRE_CLIPPED> then takes the command as the reference.  One four-cycle instruction per rotor and tick instead of two.
template <typename K> DQL_DEV float rotor_cmd(const K& s, float w2) {
  return sqrt_pos(__builtin_amdgcn_fmed3f(w2, SQRT_POS_MIN, (float)s.omax * (float)s.omax));
}
// xonly (float32, round 4): an x-axis config flies with a roll set-point of exactly 0, so B = Ry(pitch_sp) = [[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]] and the
// seven entries collapse: E10 = 0, E12 = -sin(roll), the others lose their middle term — 14 instructions instead of 24, and B is two registers, not nine.
// Compile-time in the layouts the host selects by itself (k_step's XMODE), a wave-uniform test in the others; the oracle takes the same form.
template <typename T, typename K>
DQL_DEV void attitude(const K& s, const T (&R)[9], const T (&w)[3], const T (&B)[9], T cy, T sy, T ct, T rn, T r_cmd, T thrust, T (&cmd)[4], bool xonly = false) {
  T E01, E10, E02, E20, E12, E21, E22;
  if constexpr (Fast32<T>::on) {
    const T sr = R[7] * rn, cr = R[8] * rn;             // sin, cos of the roll angle
    const T A01 = -(R[6] * sr), A02 = -(R[6] * cr);     // sin(pitch) sin(roll), sin(pitch) cos(roll)
    if (xonly) {
      const T cp = B[0], sp = B[2];
      E01 = fma_(cp, A01, -(sp * R[7])); E02 = fma_(cp, A02, -(sp * R[8]));
      E20 = fma_(sp, ct, cp * R[6]); E21 = fma_(sp, A01, cp * R[7]); E22 = fma_(sp, A02, cp * R[8]);
      const T dR0 = E21 + sr, dR1 = E02 - E20;          // E12 = -sr, E10 = 0: dR2 = -E01
      const T eW0 = fma_(-r_cmd, E02, w[0]), eW1 = fma_(r_cmd, sr, w[1]), eW2 = fma_(-r_cmd, E22, w[2]);
      const T M0 = fma_(-eW0, T(s.kW[0]), -(dR0 * (T(0.5) * T(s.kR[0]))));
      const T M1 = fma_(-eW1, T(s.kW[1]), -(dR1 * (T(0.5) * T(s.kR[1]))));
      const T M2 = fma_(-eW2, T(s.kW[2]), E01 * (T(0.5) * T(s.kR[2])));
      const T a = thrust * s.ia;
      const T w2[4] = {fma_(M2, T(s.ic), fma_(-M1, T(s.ib), a)), fma_(-M2, T(s.ic), fma_(M0, T(s.ib), a)), fma_(M2, T(s.ic), fma_(M1, T(s.ib), a)), fma_(-M2, T(s.ic), fma_(-M0, T(s.ib), a))};
#pragma unroll
      for (int i = 0; i < 4; ++i) cmd[i] = rotor_cmd(s, w2[i]);
      return;
    }
    E01 = fma_(B[0], A01, fma_(B[3], cr, B[6] * R[7]));
    E02 = fma_(B[0], A02, fma_(B[3], -sr, B[6] * R[8]));
    E10 = fma_(B[1], ct, B[7] * R[6]);
    E12 = fma_(B[1], A02, fma_(B[4], -sr, B[7] * R[8]));
    E20 = fma_(B[2], ct, B[8] * R[6]);
    E21 = fma_(B[2], A01, fma_(B[5], cr, B[8] * R[7]));
    E22 = fma_(B[2], A02, fma_(B[5], -sr, B[8] * R[8]));
  } else {
    T D[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) { D[j] = fma_(cy, B[j], -(sy * B[3 + j])); D[3 + j] = fma_(sy, B[j], cy * B[3 + j]); D[6 + j] = B[6 + j]; }
#define DQL_E(i, j) fma_(D[i], R[j], fma_(D[3 + i], R[3 + j], D[6 + i] * R[6 + j]))
    E01 = DQL_E(0, 1); E10 = DQL_E(1, 0); E02 = DQL_E(0, 2); E20 = DQL_E(2, 0); E12 = DQL_E(1, 2); E21 = DQL_E(2, 1); E22 = DQL_E(2, 2);
#undef DQL_E
  }
  if constexpr (Fast32<T>::on) {
    // e_R = (E_ji - E_ij) / 2 and M = -k_W e_W - k_R e_R: the halving is folded into the gain, (0.5 d) k == d (0.5 k) bit for bit (scaling by a
    // power of two is exact), three multiplications fewer per tick and the same value — the oracle keeps the reference's order
    const T dR0 = E21 - E12, dR1 = E02 - E20, dR2 = E10 - E01;
    const T eW0 = fma_(-r_cmd, E02, w[0]), eW1 = fma_(-r_cmd, E12, w[1]), eW2 = fma_(-r_cmd, E22, w[2]);
    const T M0 = fma_(-eW0, T(s.kW[0]), -(dR0 * (T(0.5) * T(s.kR[0]))));
    const T M1 = fma_(-eW1, T(s.kW[1]), -(dR1 * (T(0.5) * T(s.kR[1]))));
    const T M2 = fma_(-eW2, T(s.kW[2]), -(dR2 * (T(0.5) * T(s.kR[2]))));
    const T a = thrust * s.ia;
    const T w2[4] = {fma_(M2, T(s.ic), fma_(-M1, T(s.ib), a)), fma_(-M2, T(s.ic), fma_(M0, T(s.ib), a)), fma_(M2, T(s.ic), fma_(M1, T(s.ib), a)), fma_(-M2, T(s.ic), fma_(-M0, T(s.ib), a))};
#pragma unroll
    for (int i = 0; i < 4; ++i) cmd[i] = rotor_cmd(s, w2[i]);
  } else {
    const T eR0 = T(0.5) * (E21 - E12), eR1 = T(0.5) * (E02 - E20), eR2 = T(0.5) * (E10 - E01);
    const T eW0 = w[0] - r_cmd * E02, eW1 = w[1] - r_cmd * E12, eW2 = w[2] - r_cmd * E22;
    const T M0 = -(eR0 * s.kR[0]) - eW0 * s.kW[0];
    const T M1 = -(eR1 * s.kR[1]) - eW1 * s.kW[1];
    const T M2 = -(eR2 * s.kR[2]) - eW2 * s.kW[2];
    const T a = thrust * s.ia, bx = M0 * s.ib, by = M1 * s.ib, cz = M2 * s.ic;
    const T w2[4] = {a - by + cz, a + bx - cz, a + by + cz, a - bx - cz};
#pragma unroll
    for (int i = 0; i < 4; ++i) cmd[i] = sqrt_(w2[i] > T(0.0) ? w2[i] : T(0.0));
  }
}
// gazebo_motor_model.cpp:434-500 + semi-implicit Euler of one rigid body
// first-order rotor speed filter (common.h:147-183), commanded speed clipped at max_rot_velocity (gazebo_motor_model.cpp:358-364)
template <bool PRE_CLIPPED = false, typename T, typename K> DQL_DEV void rotor_filter(const K& s, Env<T>& e, const T (&cmd)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T ref = (PRE_CLIPPED && Fast32<T>::on) ? cmd[i] : clip3(cmd[i], T(0.0), T(s.omax));  // cmd = sqrt(..) >= +0: min(cmd, omax); float32 tick: rotor_cmd() did it
    if constexpr (Fast32<T>::on) {  // om + (1 - a) (ref - om): the same filter, one instruction less
      const T d = ref - e.om[i];
      // (round 5 tried max(fma(oup, d, om), fma(odn, d, om)) — the same value bit for bit since 0 < odn < oup, one instruction less and one instead of two in
      // the four-cycle class: 19.20 vs 19.18 us per period, nothing: the tick sits within 7 % of its issue cost and 1 % is the noise of code placement)
#ifdef DQL_AB_ROTOR_SELECT  // A/B builds (tools/ab_build.sh): the compare + select form
      const T c = d > T(0.0) ? T(s.oup) : T(s.odn);
      e.om[i] = fma_(c, d, e.om[i]);
#else
      // the larger of the two candidates IS the selected one (0 < odn < oup: oup d > odn d for d > 0, < for d < 0, equal at 0): two fmas and a max — one
      // instruction of the four-cycle class per rotor instead of two (compare + select), bit for bit the same value
      const T up = fma_(T(s.oup), d, e.om[i]), dn = fma_(T(s.odn), d, e.om[i]);
      if constexpr (sizeof(T) == 4) e.om[i] = __builtin_fmaxf(up, dn); else e.om[i] = up > dn ? up : dn;
#endif
    } else {
      const T a = ref > e.om[i] ? s.aup : s.adn;
      e.om[i] = fma_(a, e.om[i], (T(1.0) - a) * ref);
    }
  }
}
// forces from the CURRENT rotor speeds (gazebo_motor_model.cpp:434-500) + semi-implicit Euler of one rigid body
template <typename T, typename K> DQL_DEV void plant_step(const K& s, Env<T>& e, const T (&R)[9]) {
  const T l = s.l, h = s.h;
  const T w0 = e.w[0], w1 = e.w[1], w2 = e.w[2];
  // thrust k_f om_i^2 along body z at rotor i = (+l,0,h), (0,+l,h), (-l,0,h), (0,-l,h); drag torque -dir_i k_m T_i
  T Fbz, tx, ty, tz, S, d02, d13;
  if constexpr (Fast32<T>::on) {
    // round 4b: opposite rotors first — the sums and differences of the speeds of arm x (0, 2) and arm y (1, 3) serve the thrust torques
    // (q1 - q3 = (om1 - om3)(om1 + om3)), the drag sums and the total alike; squares by fma: 17 instructions instead of 21
    const T s02 = e.om[0] + e.om[2], s13 = e.om[1] + e.om[3];
    d02 = e.om[0] - e.om[2]; d13 = e.om[1] - e.om[3];
    S = s02 + s13;
    const T qa_ = fma_(e.om[0], e.om[0], e.om[2] * e.om[2]), qb_ = fma_(e.om[1], e.om[1], e.om[3] * e.om[3]);  // q0 + q2, q1 + q3
    Fbz = s.kf * (qa_ + qb_);
    tx = s.lkf * (d13 * s13); ty = -(s.lkf * (d02 * s02)); tz = s.kmkf * (qa_ - qb_);
  } else {
    const T q0 = e.om[0] * e.om[0], q1 = e.om[1] * e.om[1], q2 = e.om[2] * e.om[2], q3 = e.om[3] * e.om[3];
    Fbz = s.kf * ((q0 + q1) + (q2 + q3));
    tx = s.lkf * (q1 - q3); ty = s.lkf * (q2 - q0); tz = s.kmkf * ((q0 - q1) + (q2 - q3));
    S = (e.om[0] + e.om[1]) + (e.om[2] + e.om[3]); d02 = e.om[0] - e.om[2]; d13 = e.om[1] - e.om[3];
  }
  // rotor drag -|om_i| c_d v_perp,i with v_perp,i = (v_body + w x r_i) restricted to the rotor plane, summed in closed form:
  // sum_i om_i (w x r_i)_x = S w_y h - w_z l (om1 - om3),  sum_i om_i (w x r_i)_y = -S w_x h + w_z l (om0 - om2)
  const T vbx = fma_(R[0], e.v[0], fma_(R[3], e.v[1], R[6] * e.v[2]));
  const T vby = fma_(R[1], e.v[0], fma_(R[4], e.v[1], R[7] * e.v[2]));
  const T uxc = fma_(w1, h, vbx), uyc = fma_(-w0, h, vby), wzl = w2 * l;
  const T Fbx = -(s.cd * fma_(S, uxc, -(wzl * d13)));
  const T Fby = -(s.cd * fma_(S, uyc, wzl * d02));
  if constexpr (Fast32<T>::on) {  // the drag's yaw torque with -(l c_d) as one host constant
    tz = fma_(T(s.nlcd), fma_(uyc, d02, fma_(wzl, S, -(uxc * d13))), tz);
    tx = fma_(-h, Fby, tx); ty = fma_(h, Fbx, ty);
  } else {
    const T tzd = -(s.cd * fma_(uyc, d02, fma_(wzl, S, -(uxc * d13))));  // sum_i (r_i x drag_i)_z / l
    tx = fma_(-h, Fby, tx); ty = fma_(h, Fbx, ty); tz = fma_(l, tzd, tz);
  }
  tx = fma_(s.crd, Fbx, tx); ty = fma_(s.crd, Fby, ty);  // rolling moment = (c_r / c_d) * drag force
  const T Fwx = fma_(R[0], Fbx, fma_(R[1], Fby, R[2] * Fbz)), Fwy = fma_(R[3], Fbx, fma_(R[4], Fby, R[5] * Fbz)), Fwz = fma_(R[6], Fbx, fma_(R[7], Fby, R[8] * Fbz));
  if constexpr (Fast32<T>::on) {
    e.v[0] = fma_(T(s.dtm), Fwx, e.v[0]); e.v[1] = fma_(T(s.dtm), Fwy, e.v[1]); e.v[2] = fma_(T(s.dtm), Fwz, e.v[2]) - s.dtg;
  } else {
    const T ax = Fwx * s.inv_m, ay = Fwy * s.inv_m, az = Fwz * s.inv_m - s.g;
    e.v[0] = fma_(s.dt, ax, e.v[0]); e.v[1] = fma_(s.dt, ay, e.v[1]); e.v[2] = fma_(s.dt, az, e.v[2]);
  }
  e.p[0] = fma_(s.dt, e.v[0], e.p[0]); e.p[1] = fma_(s.dt, e.v[1], e.p[1]); e.p[2] = fma_(s.dt, e.v[2], e.p[2]);
  const T Iw0 = s.I[0] * w0, Iw1 = s.I[1] * w1, Iw2 = s.I[2] * w2;
  const T g0 = fma_(w1, Iw2, -(w2 * Iw1)), g1 = fma_(w2, Iw0, -(w0 * Iw2)), g2 = fma_(w0, Iw1, -(w1 * Iw0));
  if constexpr (Fast32<T>::on) {
    e.w[0] = fma_(T(s.dtI[0]), tx - g0, w0); e.w[1] = fma_(T(s.dtI[1]), ty - g1, w1);
    // round 4: (w x I w)_z = (I_y - I_x) w_x w_y is exactly 0 for a vehicle with I_x = I_y (the reference's; wave-uniform / compile-time test)
    if (T(s.I[0]) == T(s.I[1])) e.w[2] = fma_(T(s.dtI[2]), tz, w2);
    else e.w[2] = fma_(T(s.dtI[2]), tz - g2, w2);
  } else {
    e.w[0] = fma_(s.dt, (tx - g0) * s.inv_I[0], w0);
    e.w[1] = fma_(s.dt, (ty - g1) * s.inv_I[1], w1);
    e.w[2] = fma_(s.dt, (tz - g2) * s.inv_I[2], w2);
  }
  const T qw = e.q[0], qx = e.q[1], qy = e.q[2], qz = e.q[3];
  T nw, nx, ny, nz;
  if constexpr (Fast32<T>::on) {  // round 4b: the body rates scaled by dt / 2 once, every component three fmas onto the old one: 15 instructions instead of 17
    const T h0 = T(s.hdt) * e.w[0], h1 = T(s.hdt) * e.w[1], h2 = T(s.hdt) * e.w[2];
    nw = fma_(-qx, h0, fma_(-qy, h1, fma_(-qz, h2, qw)));
    nx = fma_(qw, h0, fma_(qy, h2, fma_(-qz, h1, qx)));
    ny = fma_(qw, h1, fma_(qz, h0, fma_(-qx, h2, qy)));
    nz = fma_(qw, h2, fma_(qx, h1, fma_(-qy, h0, qz)));
  } else {
    const T hdt = T(0.5) * s.dt;
    const T dw = -fma_(qx, e.w[0], fma_(qy, e.w[1], qz * e.w[2]));
    const T dxq = fma_(qw, e.w[0], fma_(qy, e.w[2], -(qz * e.w[1])));
    const T dyq = fma_(qw, e.w[1], fma_(qz, e.w[0], -(qx * e.w[2])));
    const T dzq = fma_(qw, e.w[2], fma_(qx, e.w[1], -(qy * e.w[0])));
    nw = fma_(hdt, dw, qw); nx = fma_(hdt, dxq, qx); ny = fma_(hdt, dyq, qy); nz = fma_(hdt, dzq, qz);
  }
  // renormalise with one Newton step of 1/sqrt(|q|^2) about 1: |q|^2 - 1 = O((dt |w|)^2), so the residual is O(dt^4)
  const T inv = fma_(T(-0.5), fma_(nw, nw, fma_(nx, nx, fma_(ny, ny, nz * nz))), T(1.5));
  e.q[0] = nw * inv; e.q[1] = nx * inv; e.q[2] = ny * inv; e.q[3] = nz * inv;
}
// moving_platform.py:87-127
template <typename T> DQL_DEV void platform_set(const SimK<T>& s, Env<T>& e, T sn, T cs) {
  if (s.traj == DQL_TRAJ_EIGHT) {
    e.mp_x = e.mp_r * cs; e.mp_y = e.mp_r * sn * cs;
    e.mp_u = -(e.mp_r * e.mp_w) * sn; e.mp_v = e.mp_r * e.mp_w * (cs * cs - sn * sn);
  } else {
    e.mp_x = e.mp_r * sn; e.mp_y = T(0.0);
    e.mp_u = e.mp_r * e.mp_w * cs; e.mp_v = T(0.0);
  }
}
template <typename T> DQL_DEV void platform_eval(const SimK<T>& s, Env<T>& e) {
  T sn, cs;
  det_sincos(e.mp_phase, sn, cs);
  platform_set(s, e, sn, cs);
}
// FLOAT32 STEP, ROUND 4: inside one agent period the platform's sine and cosine are evaluated ONCE, at the period's first manager tick, and carried
// to the following ticks (four or five per period) by the rotation through the constant phase step delta = omega mp_dt — four instructions per
// manager tick instead of a 45-instruction sincos.  The phase itself advances exactly as before and stays the persistent state: every period
// starts again from sincos(phase), so the recurrence never runs for more than five steps (a few ulp).  sin / cos of delta: Taylor to delta^5 /
// delta^6 for delta <= 0.25 rad (1e-8 relative; the reference platform steps 0.008 rad), det_sincos beyond (per lane).
#ifdef DQL_PLATREC_LEAN  // A/B build: sin / cos of the step recomputed at every manager tick (9 instructions) instead of held in two registers across the tick loop
template <typename T> struct PlatRec { T sn, cs; };
#else
template <typename T> struct PlatRec { T sn, cs, sd, cd; };
#endif
template <typename T> DQL_DEV void platform_step_sincos(const SimK<T>& s, const Env<T>& e, T& sd, T& cd) {
  const T d = e.mp_w * s.mp_dt;
  if (d > T(0.25) || d < T(-0.25)) det_sincos(d, sd, cd);
  else {
    const T z = d * d;
    sd = d * fma_(z, fma_(z, T(8.33333333333333322e-03), T(-1.66666666666666657e-01)), T(1.0));
    cd = fma_(z, fma_(z, fma_(z, T(-1.38888888888888894e-03), T(4.16666666666666644e-02)), T(-0.5)), T(1.0));
  }
}
template <typename T> DQL_DEV void platform_rec_begin(const SimK<T>& s, const Env<T>& e, PlatRec<T>& r) {
  det_sincos(e.mp_phase, r.sn, r.cs);
#ifndef DQL_PLATREC_LEAN
  platform_step_sincos(s, e, r.sd, r.cd);
#endif
}
// rec: the fused float32 step's per-period sine / cosine carry (null: evaluate sincos(phase) at this tick — float64, and the stand-alone operators)
template <typename T> DQL_DEV void platform_update(const SimK<T>& s, Env<T>& e, PlatRec<T>* rec = nullptr, bool first_in_period = true) {
  if (Fast32<T>::on && rec) {
    if (first_in_period) platform_rec_begin(s, e, *rec);
    platform_set(s, e, rec->sn, rec->cs);
    const T sn = rec->sn, cs = rec->cs;
#ifdef DQL_PLATREC_LEAN
    T sd, cd;
    platform_step_sincos(s, e, sd, cd);
#else
    const T sd = rec->sd, cd = rec->cd;
#endif
    rec->sn = fma_(sn, cd, cs * sd);
    rec->cs = fma_(cs, cd, -(sn * sd));
  } else platform_eval(s, e);
  T ph = fma_(e.mp_w, s.mp_dt, e.mp_phase);
  if (ph >= T(6.28318530717958623200e+00)) ph -= T(6.28318530717958623200e+00);
  e.mp_phase = ph;
}
// manager_node.py:192-214 + observation_utils.py:77-158
// scripts/manager_node.py:292-310: plant states of the two PIDs (v_z of the drone, yaw of q_drone q_platform^-1)
template <typename T> DQL_DEV void manager_states(const T (&R)[9], T cy, T sy, T vz, T& vz_state, T& yw_state) {
  vz_state = vz;
  const T A00 = fma_(cy, R[0], sy * R[3]), A01 = fma_(cy, R[1], sy * R[4]);
  const T A10 = fma_(cy, R[3], -(sy * R[0])), A11 = fma_(cy, R[4], -(sy * R[1]));
  yw_state = det_atan2(fma_(A10, cy, A11 * sy), fma_(A00, cy, A01 * sy));
}
// manager_node.py:192-214 + observation_utils.py:77-158: relative observation, acceleration estimate, contact latch; THEN the
// platform set-point of the next 10 ms
// with_noise: the tick's noise draw is applied.  Inside the fused step only the LAST manager tick of an agent period needs it: the
// noise sits on the published p / v only (the acceleration estimate runs on the clean velocity, G12), every tick overwrites the latched
// observation, and the MDP reads the latch once, at the end of the period — so the draws of the earlier ticks (Philox + two Box-Muller
// pairs each, ~1 300 instructions per period) are never consumed and are not made.  Same values, bit for bit, as drawing every tick.
template <typename T>
DQL_DEV void manager_obs(const SimK<T>& s, Env<T>& e, T cy, T sy, long long mgr_index, uint32_t k0, uint32_t k1,
                         uint32_t step_lo, uint32_t step_hi, uint32_t env_id, uint32_t mgr_in_step, bool with_noise = true, PlatRec<T>* rec = nullptr, const uint32_t* kv = nullptr) {
  const T dxw = e.mp_x - e.p[0], dyw = e.mp_y - e.p[1];
  const T dvx = e.mp_u - e.v[0], dvy = e.mp_v - e.v[1];
  const T rpx = fma_(cy, dxw, sy * dyw), rpy = fma_(cy, dyw, -(sy * dxw));
  const T rvx = fma_(cy, dvx, sy * dvy), rvy = fma_(cy, dvy, -(sy * dvx));
  T opx = rpx, opy = rpy, ovx = rvx, ovy = rvy;
  if (with_noise && s.noisy) {
    uint32_t r[4]; T n0, n1, n2, n3;
    philox4x32(step_lo, step_hi, env_id, STREAM_NOISE0 + mgr_in_step, k0, k1, r, kv);
    box_muller(r[0], r[1], n0, n1); box_muller(r[2], r[3], n2, n3);
    opx = fma_(s.noise_p, n0, opx); opy = fma_(s.noise_p, n1, opy); ovx = fma_(s.noise_v, n2, ovx); ovy = fma_(s.noise_v, n3, ovy);
  }
  T ax_ = T(0.0), ay_ = T(0.0);
  if (!(e.flags & FL_ACC_INIT)) {
    e.vf_x = rvx; if (s.two_axis) e.vf_y = rvy; e.flags |= FL_ACC_INIT;
  } else {
    T dt_;
    if (s.quirks & DQL_Q_FROZEN_ACC_REFERENCE) dt_ = (T)mgr_index * s.mgr_dt;
    else dt_ = s.mgr_dt;
    if (dt_ <= T(0.0)) dt_ = T(0.01);
    if (Fast32<T>::on && !(s.quirks & DQL_Q_FROZEN_ACC_REFERENCE)) {  // constant divisor: one multiplication (float32 tick)
      ax_ = kalman1d(e.kal_x_x, e.kal_x_P, s.kal_q, s.kal_r, (rvx - e.vf_x) * s.inv_mgr_dt, s.kal_pss, s.kal_kss, s.kal_r_zero != 0);
      if (s.two_axis) ay_ = kalman1d(e.kal_y_x, e.kal_y_P, s.kal_q, s.kal_r, (rvy - e.vf_y) * s.inv_mgr_dt, s.kal_pss, s.kal_kss, s.kal_r_zero != 0);
    } else {
      ax_ = kalman1d(e.kal_x_x, e.kal_x_P, s.kal_q, s.kal_r, (rvx - e.vf_x) / dt_, s.kal_pss, s.kal_kss, s.kal_r_zero != 0);
      if (s.two_axis) ay_ = kalman1d(e.kal_y_x, e.kal_y_P, s.kal_q, s.kal_r, (rvy - e.vf_y) / dt_, s.kal_pss, s.kal_kss, s.kal_r_zero != 0);
    }
    if (!(s.quirks & DQL_Q_FROZEN_ACC_REFERENCE)) { e.vf_x = rvx; if (s.two_axis) e.vf_y = rvy; }
  }
  e.obs_px = opx; e.obs_py = opy; e.obs_vx = ovx; e.obs_vy = ovy; e.obs_ax = ax_; e.obs_ay = ay_;
  if (e.flags & FL_CONTACT) e.flags |= FL_OBS_CONTACT; else e.flags &= ~FL_OBS_CONTACT;
  platform_update(s, e, rec, mgr_in_step == 0);
}
// platform extrapolation between manager ticks + bumper contact test
template <typename T, typename K> DQL_DEV void platform_contact(const K& s, Env<T>& e, T low_z) {
  e.mp_x = fma_(e.mp_u, s.dt, e.mp_x); e.mp_y = fma_(e.mp_v, s.dt, e.mp_y);
  // the footprint test only where some lane of the wave is low enough to touch (a real branch: a training flight descends at 0.1 m/s from
  // 4 m and ends after 20 s at the latest — it never gets there, and the tick pays two instructions instead of eight)
  bool low;
  if constexpr (Fast32<T>::on) low = e.p[2] <= low_z;  // round 4b: against the host's mp_top + bottom (low_z: s.low_z, from a register of the caller's choosing)
  else low = e.p[2] - s.bottom <= s.mp_top;
  if (__ballot(low) != 0ull) {
    asm volatile("; footprint test" ::: "memory");  // keeps the block a block: the compiler otherwise flattens it into selects again
    if (low && abs_(e.p[0] - e.mp_x) <= s.mp_hx && abs_(e.p[1] - e.mp_y) <= s.mp_hy) e.flags |= FL_CONTACT;
  }
}
template <typename T, typename K> DQL_DEV void platform_contact(const K& s, Env<T>& e) { platform_contact(s, e, T(s.low_z)); }
// B = Rx(roll_sp) Ry(pitch_sp) (attitude_controller.py:138-140), constant over one agent period
template <typename T> DQL_DEV void make_B(T pitch_sp, T roll_sp, T (&B)[9]) {
  T sp_, cp_, sr_, cr_;
  det_sincos(pitch_sp, sp_, cp_);
  // x-axis configs fly with a roll set-point of exactly 0 in every lane: det_sincos(+-0) is (+0, 1) bit for bit (fn = +-0, r = +0,
  // sin_k(+0) = +0, cos_k(+0) = 1), so the whole wave skips the second evaluation; two-axis lanes with roll != 0 take it as before
  if (roll_sp == T(0.0)) { sr_ = T(0.0); cr_ = T(1.0); }
  else det_sincos(roll_sp, sr_, cr_);
  B[0] = cp_; B[1] = T(0.0); B[2] = sp_;
  B[3] = sr_ * sp_; B[4] = cr_; B[5] = -(sr_ * cp_);
  B[6] = -(cr_ * sp_); B[7] = sr_; B[8] = cr_ * cp_;
}

// drone start coordinate along one axis from the random offset x0 and the platform coordinate (init_mode = cfg.init_uniform):
// 0 / 1  TrainingLandingEnv.reset (pkg/landing_simulation_env.py:205-209): clip(x0 + mp, mp - p_max, mp + p_max)
// 2      SimulationLandingEnv.reset (:339-343): clip(mp - x0, -p_max, p_max) — the offset is subtracted and the clip is absolute
template <typename T> DQL_DEV T place_axis(int init_mode, T x0, T mp, T p_max) {
  if (init_mode == 2) return clip(mp - x0, -p_max, p_max);
  return clip(x0 + mp, mp - p_max, mp + p_max);
}

// ---------------------------------------------------------------------------------------------
// Packed float32 physics tick (round 2): the same IEEE operations, component by component and in the same order, as the scalar
// functions above (quat_to_R, yaw_cs, pid_output x 2, attitude, plant_step, rotor_filter, platform_contact) — so the CPU oracle
// needs no change and parity stays bit-exact — but issued two at a time as v_pk_mul / v_pk_add / v_pk_fma_f32 on register PAIRS
// that are laid out for it once per agent period: quaternion (w,x) (y,z); body rates, velocity, position (0,1) + the third
// component; rotors by arm (0,2) (1,3); the two PIDs as one pair of controllers (v_z, yaw); platform (x,y) (u,v).  For a wave that is
// ALONE on its SIMD a packed instruction costs about what a scalar one does (2.4-2.6 ns against 2.2-2.45, profiles/r2_pk_variants.jsonl),
// so every pair halves its share of the step; beside a second wave it costs two issue slots and the layout loses — the host picks it
// for batches of at most one env wave per SIMD only.  Swizzles (swap, broadcast) fold into op_sel, sign flips into neg modifiers.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
DQL_DEV f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DQL_DEV f2 bc2(float a) { return f2{a, a}; }
DQL_DEV f2 swp2(f2 a) { return __builtin_shufflevector(a, a, 1, 0); }
DQL_DEV f2 lo2(f2 a) { return __builtin_shufflevector(a, a, 0, 0); }
DQL_DEV f2 hi2(f2 a) { return __builtin_shufflevector(a, a, 1, 1); }

// Identity the optimiser cannot see through: without it the pair constructors below are merged into <2 x float> accesses of Env's
// arrays, which overlap the 16-byte quad accesses of load_env / store_env only partially — the scalar-replacement pass then gives
// up on the whole struct and the compiler parks it in LDS (measured: 72 B per lane, +4 us per launch)
DQL_DEV float opq(float x) { asm("" : "+v"(x)); return x; }
DQL_DEV double opq(double x) { return x; }
DQL_DEV f2 mk2(float a, float b) { return f2{opq(a), opq(b)}; }
struct TickPk {
  f2 q_wx, q_yz, w01, v01, p01, omA, omB;  // rotors: A = (0, 1), B = (2, 3): opposite rotors sit in the same half of their pairs
  float w2, v2, p2;
  f2 pid_i, pid_x1, pid_x2, pid_y1, pid_state;  // (v_z controller, yaw controller); x1, x2, y1 = the transposed Butterworth's t1, t2, t3
  f2 mp_xy, mp_uv;
};
struct RotPk { f2 R04, R13, R26, R57, R01, R34, R67; float R8, cy, sy, ct, rn; };  // rotation matrix: symmetric partners + row pairs; yaw frame

DQL_DEV void pack_tick(const Env<float>& e, TickPk& s) {
  s.q_wx = mk2(e.q[0], e.q[1]); s.q_yz = mk2(e.q[2], e.q[3]);
  s.w01 = mk2(e.w[0], e.w[1]); s.w2 = e.w[2];
  s.v01 = mk2(e.v[0], e.v[1]); s.v2 = e.v[2];
  s.p01 = mk2(e.p[0], e.p[1]); s.p2 = e.p[2];
  s.omA = mk2(e.om[0], e.om[1]); s.omB = mk2(e.om[2], e.om[3]);
  s.pid_i = mk2(e.vz_i, e.yw_i); s.pid_x1 = mk2(e.vz_x1, e.yw_x1); s.pid_x2 = mk2(e.vz_x2, e.yw_x2);
  s.pid_y1 = mk2(e.vz_y1, e.yw_y1);
  s.pid_state = mk2(e.vz_state, e.yw_state);
  s.mp_xy = mk2(e.mp_x, e.mp_y); s.mp_uv = mk2(e.mp_u, e.mp_v);
}
DQL_DEV void unpack_tick(const TickPk& s, Env<float>& e) {
  e.q[0] = opq(s.q_wx.x); e.q[1] = opq(s.q_wx.y); e.q[2] = opq(s.q_yz.x); e.q[3] = opq(s.q_yz.y);
  e.w[0] = opq(s.w01.x); e.w[1] = opq(s.w01.y); e.w[2] = s.w2;
  e.v[0] = opq(s.v01.x); e.v[1] = opq(s.v01.y); e.v[2] = s.v2;
  e.p[0] = opq(s.p01.x); e.p[1] = opq(s.p01.y); e.p[2] = s.p2;
  e.om[0] = opq(s.omA.x); e.om[1] = opq(s.omA.y); e.om[2] = opq(s.omB.x); e.om[3] = opq(s.omB.y);
  e.vz_i = opq(s.pid_i.x); e.yw_i = opq(s.pid_i.y); e.vz_x1 = opq(s.pid_x1.x); e.yw_x1 = opq(s.pid_x1.y); e.vz_x2 = opq(s.pid_x2.x); e.yw_x2 = opq(s.pid_x2.y);
  e.vz_y1 = opq(s.pid_y1.x); e.yw_y1 = opq(s.pid_y1.y);
  e.vz_state = opq(s.pid_state.x); e.yw_state = opq(s.pid_state.y);
  e.mp_x = opq(s.mp_xy.x); e.mp_y = opq(s.mp_xy.y); e.mp_u = opq(s.mp_uv.x); e.mp_v = opq(s.mp_uv.y);
}
// quat_to_R + yaw_cs
DQL_DEV void rot_pk(const TickPk& s, RotPk& r) {
  // quat_to_R's float32 form (round 4), component by component: doubled x, y, z, then one or two fmas per entry
  const f2 xy = f2{s.q_wx.y, s.q_yz.x};
  const float w = s.q_wx.x, z = s.q_yz.y;
  const f2 xy2 = xy + xy;                                                  // (x2, y2)
  const float z2 = z + z;
  const float t = fma_(-z, z2, 1.0f);
  r.R04 = pfma(-swp2(xy), swp2(xy2), bc2(t));                              // R0 = fma(-y, y2, t), R4 = fma(-x, x2, t)
  r.R8 = fma_(-xy.x, xy2.x, fma_(-xy.y, xy2.y, 1.0f));
  const f2 wxy2 = bc2(w) * xy2;                                            // (w x2, w y2)
  const float wz2 = w * z2;
  r.R13 = pfma(lo2(xy), hi2(xy2), f2{-wz2, wz2});                          // fma(x, y2, -+wz2)
  r.R26 = pfma(lo2(xy), bc2(z2), f2{wxy2.y, -wxy2.y});                     // fma(x, z2, +-wy2)
  r.R57 = pfma(hi2(xy), bc2(z2), f2{-wxy2.x, wxy2.x});                     // fma(y, z2, -+wx2)
  r.R01 = f2{r.R04.x, r.R13.x}; r.R34 = f2{r.R13.y, r.R04.y}; r.R67 = f2{r.R26.y, r.R57.y};
  const float R0 = r.R04.x, R3 = r.R13.y;
  const float n2 = fma_(R0, R0, R3 * R3);
  const float rr = yaw_rnorm(n2);
  r.cy = R0 * rr; r.sy = R3 * rr; r.ct = n2 * rr; r.rn = rr;
}
DQL_DEV void rot_to_array(const RotPk& r, float (&R)[9]) {
  R[0] = r.R04.x; R[1] = r.R13.x; R[2] = r.R26.x; R[3] = r.R13.y; R[4] = r.R04.y; R[5] = r.R57.x; R[6] = r.R26.y; R[7] = r.R57.y; R[8] = r.R8;
}
// constants of the packed tick (register pairs), built once per agent period
struct PkK {
  f2 kp, ki, lo, hi, wind, sp;     // the two PIDs: (v_z, yaw)
  f2 kRn, kW01, I01, dtI01;        // kRn = (kR0, -kR1) / 2
};
template <typename K> DQL_DEV PkK make_pkk(const K& c) {
  PkK k;
  k.kp = mk2(c.vz_kp, c.yw_kp); k.ki = mk2(c.vz_ki, c.yw_ki); k.lo = mk2(c.vz_lo, c.yw_lo); k.hi = mk2(c.vz_hi, c.yw_hi);
  k.wind = mk2(c.vz_wind, c.yw_wind); k.sp = mk2(c.vz_sp, c.yw_sp);
  k.kRn = mk2(0.5f * c.kR[0], -(0.5f * c.kR[1])); k.kW01 = mk2(c.kW[0], c.kW[1]); k.I01 = mk2(c.I[0], c.I[1]); k.dtI01 = mk2(c.dtI[0], c.dtI[1]);
  return k;
}
// one 500 Hz physics tick after the rotation (and the manager tick, if due): both PIDs, attitude law, rotor model, rigid body,
// rotor filter, platform extrapolation + contact test.  B = Rx(roll_sp) Ry(pitch_sp) as pairs B01, B34, B67 and scalars B2, B5, B8.
template <bool XONLY, typename K>
DQL_DEV void physics_tick_pk(const K& c, const PkK& k, TickPk& s, const RotPk& r, const f2 B01, const f2 B34, const f2 B67, const float B2,
                             const float B5, const float B8, int& flags) {
  // ---- pid_output x 2 (pid.py:62-104, Kd = 0) ----
  const f2 e0 = k.sp - s.pid_state;
  const f2 ii = pfma(e0, bc2(c.dt), s.pid_i);
  s.pid_i = f2{clip3(ii.x, -k.wind.x, k.wind.x), clip3(ii.y, -k.wind.y, k.wind.y)};
  const f2 fe = pfma(bc2(c.bw_inv), e0, s.pid_x1);                           // transposed Butterworth (butterworth(), float32 form)
  s.pid_x1 = pfma(bc2(c.bw_b2), e0, s.pid_x2);
  s.pid_x2 = pfma(bc2(c.bw_inv), e0, s.pid_y1);
  if (c.bw_k2 != 0) s.pid_x2 = pfma(-bc2(c.bw_a2), fe, s.pid_x2);
  s.pid_y1 = -(bc2(c.bw_a3) * fe);
  const f2 eff = pfma(k.kp, fe, k.ki * s.pid_i);
  const float thrust = clip3(eff.x, k.lo.x, k.hi.x), r_cmd = clip3(eff.y, k.lo.y, k.hi.y);
  // ---- attitude law (attitude_controller.py:107-156): E = B^T A on the yaw-free attitude A (see attitude()) ----
  const float R6 = r.R26.y, R7 = r.R57.y;
  const float sr = R7 * r.rn, cr = r.R8 * r.rn;
  const f2 A012 = f2{-(R6 * sr), -(R6 * cr)};                                            // (A01, A02)
  const f2 crsr = f2{cr, -sr};                                                            // (A11, A12)
  const f2 R78 = f2{R7, r.R8};                                                            // (A21, A22)
  f2 hh, E02_12; float dR2, E22;
  if constexpr (XONLY) {  // B = Ry(pitch_sp): attitude(), xonly
    const f2 cp2 = lo2(B01), sp2 = bc2(B2);
    const f2 E01_02 = pfma(cp2, A012, -(sp2 * R78));                                      // fma(cp, A01, -(sp R7)), fma(cp, A02, -(sp R8))
    const f2 E21_22 = pfma(sp2, A012, cp2 * R78);                                         // fma(sp, A01, cp R7), fma(sp, A02, cp R8)
    const float E20 = fma_(B2, r.ct, B01.x * R6);
    E02_12 = f2{E01_02.y, -sr};                                                           // E12 = -sin(roll)
    hh = E02_12 - f2{E20, E21_22.x};                                                      // (dR1, -dR0) = (E02 - E20, -(E21 + sr))
    dR2 = -E01_02.x; E22 = E21_22.y;
  } else {
    const f2 E01_02 = pfma(lo2(B01), A012, pfma(lo2(B34), crsr, lo2(B67) * R78));           // (E01, E02)
    const f2 E21_22 = pfma(bc2(B2), A012, pfma(bc2(B5), crsr, bc2(B8) * R78));              // (E21, E22)
    const f2 E10_20 = pfma(f2{B01.y, B2}, bc2(r.ct), f2{B67.y, B8} * bc2(R6));              // (E10, E20)
    const float E12 = fma_(B01.y, A012.y, fma_(B34.y, -sr, B67.y * r.R8));
    E02_12 = f2{E01_02.y, E12};
    const f2 E20_21 = f2{E10_20.y, E21_22.x}, E01_10 = f2{E01_02.x, E10_20.x};
    E22 = E21_22.y;
    hh = E02_12 - E20_21;                                                     // 2 (eR1, -eR0): the halving sits in kRn (exact)
    dR2 = E01_10.y - E01_10.x;
  }
  const f2 eW01 = pfma(-bc2(r_cmd), E02_12, s.w01);
  const float eW2 = fma_(-r_cmd, E22, s.w2);
  const f2 M01 = pfma(-eW01, k.kW01, swp2(hh) * k.kRn);                     // fma(-eW0, kW0, -(eR0 kR0)), fma(-eW1, kW1, -(eR1 kR1))
  const float M2 = fma_(-eW2, (float)c.kW[2], -(dR2 * (0.5f * (float)c.kR[2])));
  const float a = thrust * c.ia;
  const f2 ibn = f2{-(float)c.ib, (float)c.ib}, icn = f2{(float)c.ic, -(float)c.ic};
  const f2 w2_A = pfma(bc2(M2), icn, pfma(swp2(M01), ibn, bc2(a)));          // rotors 0, 1: fma(+-M2, ic, fma(-M1 | +M0, ib, a))
  const f2 w2_B = pfma(bc2(M2), icn, pfma(swp2(M01), f2{(float)c.ib, -(float)c.ib}, bc2(a)));  // rotors 2, 3: fma(+-M2, ic, fma(+M1 | -M0, ib, a))
  const f2 cmdA = f2{rotor_cmd(c, w2_A.x), rotor_cmd(c, w2_A.y)};
  const f2 cmdB = f2{rotor_cmd(c, w2_B.x), rotor_cmd(c, w2_B.y)};
  // ---- rotor forces from the CURRENT rotor speeds + rigid body (gazebo_motor_model.cpp:434-500) ----
  const float l = c.l, h = c.h;
  // (plant_step, float32 form: sums and differences of opposite rotors' speeds first)
  const f2 so = s.omA + s.omB, dd = s.omA - s.omB;                           // (s02, s13), (d02, d13)
  const float S = so.x + so.y, d02 = dd.x, d13 = dd.y;
  const f2 qab = pfma(s.omA, s.omA, s.omB * s.omB);                          // (q0 + q2, q1 + q3)
  const float Fbz = c.kf * (qab.x + qab.y);
  float tz = c.kmkf * (qab.x - qab.y);
  const f2 ds = dd * so;                                                     // (d02 s02, d13 s13)
  f2 txy = bc2(c.lkf) * f2{ds.y, -ds.x};                                     // lkf (d13 s13), -(lkf (d02 s02))
  const f2 vb = pfma(r.R01, lo2(s.v01), pfma(r.R34, hi2(s.v01), r.R67 * bc2(s.v2)));
  const f2 u = pfma(f2{s.w01.y, -s.w01.x}, bc2(h), vb);                     // (uxc, uyc)
  const float wzl = s.w2 * l;
  const f2 md = bc2(wzl) * swp2(dd);                                         // wzl (d13, d02)
  const f2 F = -(bc2(c.cd) * pfma(bc2(S), u, f2{-md.x, md.y}));             // (Fbx, Fby)
  tz = fma_((float)c.nlcd, fma_(u.y, d02, fma_(wzl, S, -(u.x * d13))), tz);
  txy = pfma(f2{-h, h}, swp2(F), txy);
  txy = pfma(bc2(c.crd), F, txy);
  const float Fwx = fma_(r.R04.x, F.x, fma_(r.R13.x, F.y, r.R26.x * Fbz));
  const float Fwy = fma_(r.R13.y, F.x, fma_(r.R04.y, F.y, r.R57.x * Fbz));
  const float Fwz = fma_(r.R26.y, F.x, fma_(r.R57.y, F.y, r.R8 * Fbz));
  s.v01 = pfma(bc2((float)c.dtm), f2{Fwx, Fwy}, s.v01); s.v2 = fma_((float)c.dtm, Fwz, s.v2) - c.dtg;
  s.p01 = pfma(bc2(c.dt), s.v01, s.p01); s.p2 = fma_(c.dt, s.v2, s.p2);
  const float w0 = s.w01.x, w1 = s.w01.y, w2 = s.w2;
  const f2 Iw01 = k.I01 * s.w01; const float Iw2 = c.I[2] * w2;
  const float g0 = fma_(w1, Iw2, -(w2 * Iw01.y)), g1 = fma_(w2, Iw01.x, -(w0 * Iw2));
  s.w01 = pfma(k.dtI01, txy - f2{g0, g1}, s.w01);
  if ((float)c.I[0] == (float)c.I[1]) s.w2 = fma_((float)c.dtI[2], tz, w2);  // (w x I w)_z = 0 for I_x = I_y (plant_step)
  else s.w2 = fma_((float)c.dtI[2], tz - fma_(w0, Iw01.y, -(w1 * Iw01.x)), w2);
  {
    const float qw = s.q_wx.x, qx = s.q_wx.y, qy = s.q_yz.x, qz = s.q_yz.y;
    const f2 h01 = bc2((float)c.hdt) * s.w01; const float h2 = (float)c.hdt * s.w2;  // (plant_step, float32 form)
    const float h0 = h01.x, h1 = h01.y;
    // scalar fmas: the operands change partner at every level (a pair per level would have to be assembled by moves)
    const float nw = fma_(-qx, h0, fma_(-qy, h1, fma_(-qz, h2, qw)));
    const float nx = fma_(qw, h0, fma_(qy, h2, fma_(-qz, h1, qx)));
    const float ny = fma_(qw, h1, fma_(qz, h0, fma_(-qx, h2, qy)));
    const float nz = fma_(qw, h2, fma_(qx, h1, fma_(-qy, h0, qz)));
    const float inv = fma_(-0.5f, fma_(nw, nw, fma_(nx, nx, fma_(ny, ny, nz * nz))), 1.5f);
    s.q_wx = f2{nw * inv, nx * inv}; s.q_yz = f2{ny * inv, nz * inv};
  }
  // ---- first-order rotor speed filter (common.h:147-183), speed limit (gazebo_motor_model.cpp:358-364) ----
  {
    const f2 dA = cmdA - s.omA, dB = cmdB - s.omB;  // rotor_cmd() clamped at omax already
    const f2 cA = f2{dA.x > 0.0f ? (float)c.oup : (float)c.odn, dA.y > 0.0f ? (float)c.oup : (float)c.odn};
    const f2 cB = f2{dB.x > 0.0f ? (float)c.oup : (float)c.odn, dB.y > 0.0f ? (float)c.oup : (float)c.odn};
    s.omA = pfma(cA, dA, s.omA);
    s.omB = pfma(cB, dB, s.omB);
  }
  // ---- platform extrapolation between manager ticks + bumper contact test ----
  s.mp_xy = pfma(s.mp_uv, bc2(c.dt), s.mp_xy);
  const bool low = s.p2 <= (float)c.low_z;
  if (__ballot(low) != 0ull) {  // see platform_contact
    asm volatile("; footprint test" ::: "memory");
    const f2 dxy = s.p01 - s.mp_xy;
    if (low && abs_(dxy.x) <= c.mp_hx && abs_(dxy.y) <= c.mp_hy) flags |= FL_CONTACT;
  }
}

// double -> int64 fixed point, round to nearest even: llrint() for |x| <= 2^50, saturating beyond (a TD target of 2^24 = 16.7 M in units of reward; the
// accumulators would overflow long before).  The 2^52 trick: x + 1.5 * 2^52 lands in [2^52, 2^53), where one ulp is 1, so the mantissa IS the rounded
// integer and the difference of the bit patterns its value — 2 double-precision clamps, 1 addition, 1 64-bit subtraction instead of the ~25
// instructions of a generic double -> int64 conversion, twice per env-step.  The oracle saturates the same way (oracle/dql_oracle.c fx_round).
DQL_DEV long long fx_round(double x) {
  x = __builtin_fmin(__builtin_fmax(x, -0x1p50), 0x1p50);
  const double t = x + 0x1.8p52;
  return __double_as_longlong(t) - __double_as_longlong(0x1.8p52);
}

struct StepOut {  // what one env contributes to the shared tables / counters this period
  long long target_fx;  // TD target, fixed point (DQL_TARGET_FRAC_BITS)
  long long target_y_fx;
  long long reward_fx;
  int cell, cell_y;     // table * N_CELLS + idx*3+action (table 0 = Q_table_a, 1 = Q_table_b), or -1
  int decision, done;
  QRow next;            // acting-table row of the state the period ended in: the bootstrap's operand here, the next period's greedy row
};

struct PeriodCtx {
  uint32_t k0, k1, step_lo, step_hi;
  int prev_idx, prev_idy, action, action_y;
  bool is_reset;
  bool coin, coin_y;  // Double Q-learning: the table this period's transition updates (bit 31 of the action stream's third word)
};
// Start of an agent period: reset placement (landing_simulation_env.py:167-243) or eps-greedy guess + set-point update
// (double_q_learning.py:110-117, mdp.py:543-560).  TabPtr: pointer to the (read-only) acting Q tables.
template <typename T, typename TabPtr>
DQL_DEV PeriodCtx period_begin(const SimK<T>& s, Env<T>& e, const QRow& qx, TabPtr qa, TabPtr qb, int mode, uint32_t eps_thr, int ext_action, uint64_t seed,
                               uint32_t env_id, long long step_index, const uint32_t* kv = nullptr) {
  PeriodCtx c;
  c.k0 = (uint32_t)seed; c.k1 = (uint32_t)(seed >> 32); c.step_lo = (uint32_t)step_index; c.step_hi = (uint32_t)((uint64_t)step_index >> 32);
  uint32_t r[4];
  philox4x32(c.step_lo, c.step_hi, env_id, STREAM_ACTION, c.k0, c.k1, r, kv);
  c.is_reset = (e.flags & FL_DONE) != 0;
  c.prev_idx = e.idx_x; c.prev_idy = e.idx_y;
  const bool two = s.two_axis != 0;
  int action = 2, action_y = 2;
  uint32_t r2[4] = {0u, 0u, 0u, 0u};
  if (two) philox4x32(c.step_lo, c.step_hi, env_id, STREAM_ACTION + 1u, c.k0, c.k1, r2, kv);
  if (c.is_reset) {
    e.step_count = 0; e.cur_check = 0; e.code = DQL_NON_TERMINAL; e.cum_x = T(0.0); e.cum_y = T(0.0);
    e.pitch_sp = T(0.0); e.roll_sp = T(0.0);
    if (!(s.quirks & DQL_Q_SHAPING_SURVIVES_RESET)) { e.shp_p = T(0.0); e.shp_v = T(0.0); e.shp_a = T(0.0); e.shpy_p = T(0.0); e.shpy_v = T(0.0); e.shpy_a = T(0.0); }
    T x0;
    if (s.working == 0 && !s.init_uniform) { T n0, n1; box_muller(r[2], r[3], n0, n1); x0 = s.init_sigma * n0; }
    else x0 = fma_(T(2.0) * u24<T>(r[2]), s.p_max, -s.p_max);
    e.p[0] = place_axis(s.init_uniform, x0, e.mp_x, s.p_max);
    e.p[1] = T(0.0); e.p[2] = s.z_init;
    if (two) {
      T y0;
      if (s.working == 0 && !s.init_uniform) { T n0, n1; box_muller(r2[2], r2[3], n0, n1); y0 = s.init_sigma * n0; }
      else y0 = fma_(T(2.0) * u24<T>(r2[2]), s.p_max, -s.p_max);
      e.p[1] = place_axis(s.init_uniform, y0, e.mp_y, s.p_max);
    }
    e.v[0] = e.v[1] = e.v[2] = T(0.0); e.w[0] = e.w[1] = e.w[2] = T(0.0);
    e.q[0] = T(1.0); e.q[1] = e.q[2] = e.q[3] = T(0.0);
    e.flags &= ~(FL_DONE | FL_CONTACT | FL_OBS_CONTACT);
    e.flags |= FL_WAS_RESET;
  } else {
    e.flags &= ~FL_WAS_RESET;
    if (mode == MODE_EXTERNAL) { action = ext_action & 3; action_y = two ? (ext_action >> 2) & 3 : 2; }
    else {
      const int greedy = agent_predict(qx);  // row of prev_idx, requested together with the env state
      // u24(r) < eps with u24(r) = (r >> 8) 2^-24: the same comparison among integers, (r >> 8) < ceil(eps 2^24) (host: eps_threshold)
      const bool explore = (mode == MODE_TRAIN) && ((r[0] >> 8) < eps_thr);
      action = explore ? (int)(((uint64_t)r[1] * 3u) >> 32) : greedy;
      if (two) {
        const int greedy_y = agent_predict(qa, qb, c.prev_idy < 0 ? 0 : c.prev_idy);
        const bool explore_y = (mode == MODE_TRAIN) && ((r2[0] >> 8) < eps_thr);
        action_y = explore_y ? (int)(((uint64_t)r2[1] * 3u) >> 32) : greedy_y;
      }
    }
    e.pitch_sp = continuous_action(s, e.pitch_sp, action);
    if (two) e.roll_sp = -continuous_action(s, -e.roll_sp, action_y);  // theta_y = -roll
  }
  c.action = action; c.action_y = action_y;
  c.coin = (r[2] >> 31) != 0; c.coin_y = (r2[2] >> 31) != 0;
  e.action = action | (two ? action_y << 2 : 0);
  return c;
}
// End of an agent period: fresh Euler angles, discretise / check / reward (mdp.py:257-541), TD target (double_q_learning.py:136-145)
// SCALAR_MDP: fetch the MDP constants with scalar loads (constant address space) instead of twelve vector loads per lane and period.  An A/B
// of the three choices inside ONE run (all / none / this) shows no time difference beyond 0.5 % at 4 096 ... 1 M envs — differences between
// runs on different boxes are 2-3 % and had looked like an effect; what it does buy is registers: 14 VGPRs less in the multi-wave layouts, and
// the 128-VGPR variant's scratch 164 -> 92 B per lane.  The lone-wave layouts (registers to spare) keep the plain pointer.
// MDP_SRC: where the MDP constants come from — MDP_VECTOR / MDP_SCALAR: the MdpK buffer, with vector or scalar loads; MDP_LITERAL: LitM + MdpRun
enum { MDP_VECTOR = 0, MDP_SCALAR = 1, MDP_LITERAL = 2 };
template <typename T, typename M, typename TabPtr>
DQL_DEV StepOut period_end_with(const SimK<T>& s, const M& m, Env<T>& e, const PeriodCtx& c, TabPtr qa, TabPtr qb, int mode);
template <int MDP_SRC, typename T, typename TabPtr>
DQL_DEV StepOut period_end(const SimK<T>& s, const MdpK<T> DQL_CONST_AS* mp, const MdpRun<T>& mr, Env<T>& e, const PeriodCtx& c, TabPtr qa, TabPtr qb, int mode) {
  if constexpr (MDP_SRC == MDP_LITERAL && sizeof(T) == 4) {
    const LitM m{mr.timeout_steps, mr.gamma, s.working, mr.goal_logic, s.quirks};
    return period_end_with(s, m, e, c, qa, qb, mode);
  } else {
    asm volatile("" ::: "memory");  // keep the MdpK scalar loads below the tick loop
    MdpK<T> m;
    if constexpr (MDP_SRC == MDP_SCALAR) __builtin_memcpy(&m, mp, sizeof(m));
    else m = *(const MdpK<T>*)mp;
    return period_end_with(s, m, e, c, qa, qb, mode);
  }
}
template <typename T, typename M, typename TabPtr>
DQL_DEV StepOut period_end_with(const SimK<T>& s, const M& m, Env<T>& e, const PeriodCtx& c, TabPtr qa, TabPtr qb, int mode) {
  StepOut out; out.cell = -1; out.cell_y = -1; out.decision = 0; out.done = 0; out.target_fx = 0; out.target_y_fx = 0; out.reward_fx = 0;
  const bool two = s.two_axis != 0;
  const int prev_idx = c.prev_idx, prev_idy = c.prev_idy;
  T R[9];
  DQL_SECTION("end_pitch");
  quat_to_R(e.q, R);
  int idx, idy = -1;
  Bins bnx{0, 0, 0}, bny{0, 0, 0};
#ifdef DQL_AB_NO_TANBIN  // A/B builds (tools/ab_build.sh): timing only, no parity
  constexpr bool TANBIN = false;
#else
  constexpr bool TANBIN = Fast32<T>::on;
#endif
  if constexpr (TANBIN) {  // the angle bins straight from the rotation matrix (angle_bin_from_tangent): pitch = atan2(-R20, sqrt(R00^2 + R10^2)), roll = atan2(R21, R22)
    const int bin_x = angle_bin_from_tangent(m, -R[6], fma_(R[0], R[0], R[3] * R[3]), true);
    DQL_SECTION("end_discretise");
    idx = discretise_impl<true>(m, e.obs_px, e.obs_vx, e.obs_ax, T(0.0), bin_x, &bnx);
    if (two) idy = discretise_impl<true>(m, e.obs_py, e.obs_vy, e.obs_ay, T(0.0), angle_bin_from_tangent(m, -R[7], R[8] * R[8], R[8] > T(0.0)), &bny);
  } else {
    const T cyy = sqrt_(fma_(R[0], R[0], R[3] * R[3]));
    const T pitch = det_atan2(-R[6], cyy);
    DQL_SECTION("end_discretise");
    idx = discretise_impl<false>(m, e.obs_px, e.obs_vx, e.obs_ax, pitch, 0, &bnx);
    if (two) { const T roll = det_atan2(R[7], R[8]); idy = discretise_impl<false>(m, e.obs_py, e.obs_vy, e.obs_ay, -roll, 0, &bny); }
  }
  if (idx < 0) { idx = 0; bnx = Bins{0, 0, 0}; }
  e.idx_x = idx;
  if (two) {
    if (idy < 0) { idy = 0; bny = Bins{0, 0, 0}; }
    e.idx_y = idy;
  }
  const int prev_k = e.bin_k, prev_p = e.bin_p, prev_ky = e.bin_ky, prev_py = e.bin_py;  // of prev_idx / prev_idy
  e.bin_k = bnx.k; e.bin_p = bnx.p; e.bin_ky = bny.k; e.bin_py = bny.p;
  e.reward = T(0.0);
  // both tables' row of the new state, requested as soon as the index exists: check / reward below run while it travels, the TD target
  // takes it from registers, and so does the NEXT period's greedy choice when the env stays in registers (periods_per_launch > 1)
  DQL_SECTION("end_qrow");
  out.next = load_qrow(qa, qb, idx);
  DQL_MARK_T(e, 41);
  if (c.is_reset) return out;
  DQL_SECTION("end_check");
  const bool contact = (e.flags & FL_OBS_CONTACT) != 0;
  e.code = mdp_check(m, e.step_count, e.cur_check, e.code, prev_idx, idx, contact, e.obs_px, e.obs_py, e.p[2], two, prev_idy, idy, &bnx, prev_k, &bny, prev_ky);
  DQL_SECTION("end_reward");
  const T rew = mdp_reward(m, e.shp_p, e.shp_v, e.shp_a, e.cum_x, e.code, idx, e.obs_px, e.obs_vx, e.pitch_sp, &bnx);
  T rew_y = T(0.0);
  if (two) rew_y = mdp_reward(m, e.shpy_p, e.shpy_v, e.shpy_a, e.cum_y, e.code, idy, e.obs_py, e.obs_vy, -e.roll_sp, &bny);
  e.reward = two ? rew + rew_y : rew;
  DQL_MARK_T(e, 42);
  DQL_SECTION("end_target");
  const bool done = e.code <= DQL_TERMINAL_TIMEOUT;
  if (done) e.flags |= FL_DONE;
  out.decision = 1; out.done = done ? 1 : 0;
  out.reward_fx = fx_round((double)rew * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (two) out.reward_fx += fx_round((double)rew_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (mode == MODE_TRAIN) {
    // Reference (B1/B2, DQL_Q_UPDATE_TABLE_A_ONLY): always Q_table_a, valued by itself.  Otherwise Double Q-learning as the paper
    // has it: the coin picks the table to update, the OTHER table values the picked table's greedy action at s'.
    const bool dbl = !(s.quirks & DQL_Q_UPDATE_TABLE_A_ONLY);
    {
      const bool sel_b = dbl && c.coin;
      const double a0 = out.next.a0, a1 = out.next.a1, a2 = out.next.a2;
      double s0 = a0, s1 = a1, s2 = a2, v0 = a0, v1 = a1, v2 = a2;
      if (dbl) {
        const double b0 = out.next.b0, b1 = out.next.b1, b2 = out.next.b2;
        s0 = sel_b ? b0 : a0; s1 = sel_b ? b1 : a1; s2 = sel_b ? b2 : a2;
        v0 = sel_b ? a0 : b0; v1 = sel_b ? a1 : b1; v2 = sel_b ? a2 : b2;
      }
      const int b = argmax3(s0, s1, s2);
      const double boot = b == 0 ? v0 : (b == 1 ? v1 : v2);
      int mask;
      if (s.quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask = prev_p != bnx.p;
      else mask = !done;
      const double target = (double)rew + (m.gamma * boot) * (double)mask;
      out.cell = prev_idx * 3 + c.action + (sel_b ? DQL_N_CELLS : 0);
      out.target_fx = fx_round(target * (double)(1ll << DQL_TARGET_FRAC_BITS));
    }
    if (two) {  // the y transition updates the same shared tables
      const bool sel_b = dbl && c.coin_y;
      const double a0 = qa[idy * 3], a1 = qa[idy * 3 + 1], a2 = qa[idy * 3 + 2];
      double s0 = a0, s1 = a1, s2 = a2, v0 = a0, v1 = a1, v2 = a2;
      if (dbl) {
        const double b0 = qb[idy * 3], b1 = qb[idy * 3 + 1], b2 = qb[idy * 3 + 2];
        s0 = sel_b ? b0 : a0; s1 = sel_b ? b1 : a1; s2 = sel_b ? b2 : a2;
        v0 = sel_b ? a0 : b0; v1 = sel_b ? a1 : b1; v2 = sel_b ? a2 : b2;
      }
      const int by = argmax3(s0, s1, s2);
      const double boot_y = by == 0 ? v0 : (by == 1 ? v1 : v2);
      int mask_y;
      if (s.quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask_y = prev_py != bny.p;
      else mask_y = !done;
      const double target_y = (double)rew_y + (m.gamma * boot_y) * (double)mask_y;
      out.cell_y = prev_idy * 3 + c.action_y + (sel_b ? DQL_N_CELLS : 0);
      out.target_y_fx = fx_round(target_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
    }
  }
  return out;
}

// Unroll factor of the plain tick loop.  float64: 2 (fewer loop-carried moves; 63.2 vs 65.0 us per period at 131 072 envs).  float32: that
// held in rounds 1 - 2 (-9 % at 1 M envs) and stopped holding with the shorter round-3 tick: NOT unrolling is 5 - 6 % faster wherever
// the plain / literal loop runs (in-run A/B, config 4 flags: 131 072 envs 24.7 vs 26.1 us, 262 144: 42.5 vs 45.0, 1 M: 150.2 vs 160.2;
// unrolling by 3, 5, 7, 11 lands between the two)
#ifndef DQL_TICK_UNROLL_F32
#define DQL_TICK_UNROLL_F32 1
#endif
#ifndef DQL_TICK_UNROLL_F64
#define DQL_TICK_UNROLL_F64 2
#endif
template <typename T> struct TickUnroll { static constexpr int n = sizeof(T) == 4 ? DQL_TICK_UNROLL_F32 : DQL_TICK_UNROLL_F64; };
#ifndef DQL_GROUP
#define DQL_GROUP 5  // manager_div of the reference: 500 Hz physics / 100 Hz observation
#endif
// One agent period of one env in one lane.  HOT: hold the per-tick constants in VGPRs (small batches: one wave per SIMD,
// registers are free and every avoided v_readlane shortens the dependency-bound stream; at full occupancy it costs a wave).
// TICK: 0 plain loop on SGPR constants, 1 per-tick constants in VGPRs + the loop laid out per manager period, 2 the same with the
// packed float32 tick (physics_tick_pk), 3 plain loop with the reference vehicle's constants as literals (LitK)
// 4: the packed tick (2) with the reference MDP's constants as literals at the period's end (LitM) — what the host picks for small batches when the MDP is the reference's
enum { TICK_PLAIN = 0, TICK_LONE = 1, TICK_PACKED = 2, TICK_LIT = 3, TICK_PACKED_LITM = 4 };
constexpr bool tick_is_packed(int t) { return t == TICK_PACKED || t == TICK_PACKED_LITM; }
template <int TICK, typename T> struct TickK { static DQL_DEV const SimK<T>& get(const SimK<T>& s) { return s; } };
template <> struct TickK<TICK_LONE, float> { static DQL_DEV HotK<float> get(const SimK<float>& s) { return make_hot(s); } };
template <> struct TickK<TICK_PACKED, float> { static DQL_DEV HotK<float> get(const SimK<float>& s) { return make_hot(s); } };
template <> struct TickK<TICK_PACKED_LITM, float> { static DQL_DEV HotK<float> get(const SimK<float>& s) { return make_hot(s); } };
template <> struct TickK<TICK_LIT, float> { static DQL_DEV LitK get(const SimK<float>& s) { return LitK{s.vz_sp, s.yw_sp}; } };
// The tick constants in the form the layout wants them (SGPR struct, VGPR copies, literals) + the packed pairs: made ONCE per launch,
// outside the loop over the launch's agent periods
template <int TICK, typename T> struct TickConsts {
  decltype(TickK<TICK, T>::get(*(const SimK<T>*)nullptr)) h;
  PkK pk;
  DQL_DEV explicit TickConsts(const SimK<T>& s) : h(TickK<TICK, T>::get(s)) {
    if constexpr (sizeof(T) == 4 && tick_is_packed(TICK)) pk = make_pkk(h);
  }
};
// XMODE: what the kernel knows about the config's axes at compile time.  X_TWO: a two-axis config (generic attitude law); X_ONLY: an x-axis config
// (roll set-point exactly 0: attitude()'s xonly form, and the y-axis state is dead code — the caller passes a SimK whose two_axis is the
// constant 0); X_RUNTIME: decided by s.two_axis (wave-uniform), both forms in the code — the layouts the host does not pick by itself.
enum { X_TWO = 0, X_ONLY = 1, X_RUNTIME = 2 };
template <int TICK, int XMODE, typename T, typename TabPtr>
DQL_DEV StepOut agent_period(const SimK<T>& s_in, const TickConsts<TICK, T>& tc, const MdpK<T> DQL_CONST_AS* mp, const MdpRun<T>& mr, Env<T>& e, const QRow& qx, TabPtr qa, TabPtr qb, int mode, uint32_t eps_thr,
                             int ext_action, uint64_t seed, uint32_t env_id, long long step_index, long long mgr0, int sched, unsigned prio_role = 0u, const uint32_t* kv = nullptr) {
  SimK<T> s = s_in;
#ifndef DQL_AB_NO_OPAQUE_FLAGS  // A/B builds (tools/ab_build.sh)
  // Every wave-uniform condition on the run's mode, quirk bits, working level and placement rule is loop-invariant, so the compiler evaluates each ONCE before
  // the period loop and keeps it as a 64-bit lane mask — a dozen SGPR pairs the kernel does not have: they were spilled into VGPR lanes and read back at every use
  // (two v_readlane + a hazard wait each).  Made opaque here, once per period, the conditions are re-derived where they are used: a scalar compare each.
  // (the packed layout — batches of at most one wave per SIMD — is indifferent: 4 096 envs +0.3 %, 32 768 / 65 536 -0.3 %: left alone)
  if constexpr (sizeof(T) == 4 && (TICK == TICK_LIT || TICK == TICK_PLAIN))
    asm volatile("" : "+s"(s.quirks), "+s"(s.working), "+s"(s.init_uniform), "+s"(mode));
#endif
  DQL_SECTION("period_begin");
  const PeriodCtx c = period_begin(s, e, qx, qa, qb, mode, eps_thr, ext_action, seed, env_id, step_index, kv);
  T B[9];
  DQL_SECTION("make_B");
  make_B(e.pitch_sp, e.roll_sp, B);
  DQL_SECTION("tick_setup");
  constexpr bool HOT = TICK == TICK_LONE || tick_is_packed(TICK);
  const auto& h = tc.h;
  // The literal layout's constants are instruction literals — which gfx9's three-operand encodings (v_med3, v_fma with an inline constant) cannot carry: for the yaw
  // PID's clamp bounds and the yaw frame's 0.375 the compiler emitted an s_mov of the literal in front of EVERY use, three scalar instructions per physics tick.
  // Pinned to SGPRs here (opaque: nothing to rematerialise) they stay scalar operands — in VGPRs they cost more than the s_movs (three VGPR sources per v_med3).
  T yw_lo = T(h.yw_lo), yw_hi = T(h.yw_hi), yw_wind = T(h.yw_wind), c375 = T(0.375), low_z = T(h.low_z);
#ifndef DQL_AB_NO_TICK_SREGS  // A/B builds (tools/ab_build.sh)
  if constexpr (sizeof(T) == 4 && TICK == TICK_LIT) {
    asm volatile("" : "+s"(yw_wind), "+s"(yw_hi), "+s"(c375));
#ifndef DQL_AB_NO_LOWZ_SREG
    asm volatile("" : "+s"(low_z));  // (the contact test's height: the literal was moved into a scalar register in front of the compare at every tick)
#endif
    if (T(h.yw_lo) == -T(h.yw_hi)) yw_lo = -yw_hi;
  }
#endif
  DQL_MARK_T(e, 3);
  DQL_PHASE(e, 1);
  T R[9], cy, sy, ct = T(1.0), rn = T(1.0);
  uint32_t mgr_in_step = 0;
  PlatRec<T> prec;                      // platform sine / cosine carried between the manager ticks of this period (float32: platform_update)
  prec = PlatRec<T>{};
  // the period's tick schedule, derived on the host from the global tick count (make_step_args; wave-uniform): its physics ticks, the ticks since
  // the last 100 Hz manager tick, the index of the next manager tick, and the period's LAST manager tick — the only one whose observation noise
  // is ever read (manager_obs)
  const int n_ticks = sched & 0xff;
  int phase = (sched >> 8) & 0xff;
  long long mgr_index = mgr0;
  const uint32_t last_mgr = (uint32_t)(sched >> 16);
  auto manager_tick = [&]() {
    DQL_SECTION("manager");
    DQL_PHASE(e, 2);
#ifdef DQL_PRIO_TIME  // A/B: issue priority by wall-clock slice instead of by period (k_step)
    { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();
      if ((((unsigned)(t_ >> DQL_PRIO_TIME)) ^ prio_role) & 1u) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0"); }
#endif
#ifdef DQL_PRIO_MGR   // A/B: by manager-tick parity
    if ((((unsigned)mgr_index) ^ prio_role) & 1u) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0");
#endif
    manager_states(R, cy, sy, e.v[2], e.vz_state, e.yw_state);
    manager_obs(s, e, cy, sy, mgr_index, c.k0, c.k1, c.step_lo, c.step_hi, env_id, mgr_in_step, mgr_in_step == last_mgr, &prec, kv);
    ++mgr_in_step; ++mgr_index;
    DQL_PHASE(e, 3);
  };
  auto control_and_plant = [&]() {
    DQL_SECTION("pid");
    const T thrust = pid_output(h, h.vz_kp, h.vz_ki, h.vz_lo, h.vz_hi, h.vz_wind, h.vz_sp, e.vz_state, e.vz_i, e.vz_x1, e.vz_x2, e.vz_y1, e.vz_y2, e.vz_y3);
    const T r_cmd = pid_output(h, h.yw_kp, h.yw_ki, yw_lo, yw_hi, yw_wind, h.yw_sp, e.yw_state, e.yw_i, e.yw_x1, e.yw_x2, e.yw_y1, e.yw_y2, e.yw_y3);
    T cmd[4];
    DQL_SECTION("attitude");
    attitude(h, R, e.w, B, cy, sy, ct, rn, r_cmd, thrust, cmd, XMODE == X_ONLY || (XMODE == X_RUNTIME && s.two_axis == 0));
    DQL_SECTION("motor_body");
    plant_step(h, e, R);
    rotor_filter<true>(h, e, cmd);
    DQL_SECTION("platform_contact");
    platform_contact(h, e, low_z);
  };
  if constexpr (sizeof(T) == 4 && tick_is_packed(TICK)) {
    // float32: the packed tick (physics_tick_pk): the state the 500 Hz loop touches lives in register pairs for the whole period
    TickPk ts;
    pack_tick(e, ts);
    const PkK& pk = tc.pk;
    const f2 B01 = f2{B[0], B[1]}, B34 = f2{B[3], B[4]}, B67 = f2{B[6], B[7]};
    const float B2 = B[2], B5 = B[5], B8 = B[8];
    RotPk rp;
    auto tick_pk = [&](bool mgr) {
      DQL_SECTION("rot");
      rot_pk(ts, rp);
      if (mgr) {
        e.p[0] = opq(ts.p01.x); e.p[1] = opq(ts.p01.y); e.p[2] = ts.p2; e.v[0] = opq(ts.v01.x); e.v[1] = opq(ts.v01.y); e.v[2] = ts.v2;
        e.mp_x = opq(ts.mp_xy.x); e.mp_y = opq(ts.mp_xy.y); e.mp_u = opq(ts.mp_uv.x); e.mp_v = opq(ts.mp_uv.y);
        {
          DQL_SECTION("manager");
          DQL_PHASE(e, 2);
          float Rm[9];  // scoped: a long-lived array would be demoted to LDS by the compiler
          rot_to_array(rp, Rm);
          manager_states(Rm, rp.cy, rp.sy, e.v[2], e.vz_state, e.yw_state);
          manager_obs(s, e, rp.cy, rp.sy, mgr_index, c.k0, c.k1, c.step_lo, c.step_hi, env_id, mgr_in_step, mgr_in_step == last_mgr, &prec, kv);
          ++mgr_in_step; ++mgr_index;
          DQL_PHASE(e, 3);
        }
        ts.pid_state = mk2(e.vz_state, e.yw_state); ts.mp_xy = mk2(e.mp_x, e.mp_y); ts.mp_uv = mk2(e.mp_u, e.mp_v);
      }
      DQL_SECTION("tick_pk");
      if constexpr (XMODE == X_RUNTIME) {
        if (s.two_axis == 0) physics_tick_pk<true>(h, pk, ts, rp, B01, B34, B67, B2, B5, B8, e.flags);
        else physics_tick_pk<false>(h, pk, ts, rp, B01, B34, B67, B2, B5, B8, e.flags);
      } else physics_tick_pk<XMODE == X_ONLY>(h, pk, ts, rp, B01, B34, B67, B2, B5, B8, e.flags);
    };
    if constexpr (!HOT) {
#pragma unroll TickUnroll<T>::n
      for (int i = 0; i < n_ticks; ++i) {
        tick_pk(phase == 0);
        phase = (phase + 1 == s.div) ? 0 : phase + 1;
      }
    } else {
      int left = n_ticks;
      for (;;) {
        while (left > 0 && !(s.div == DQL_GROUP && phase == 0 && left >= DQL_GROUP)) {
          tick_pk(phase == 0);
          phase = (phase + 1 == s.div) ? 0 : phase + 1;
          --left;
        }
        if (left == 0) break;
        do {
          tick_pk(true);
#pragma unroll
          for (int k = 1; k < DQL_GROUP; ++k) tick_pk(false);
          left -= DQL_GROUP;
        } while (left >= DQL_GROUP);
      }
    }
    unpack_tick(ts, e);
  } else if constexpr (!(HOT && sizeof(T) == 4)) {
    // big batches (several waves per SIMD, registers decide the occupancy): the plain loop
    // (physics ticks to go until the next manager tick, counted down: a compare + branch per tick; the phase counted up and wrapped cost an add, a compare,
    //  a select and the compare + branch)
    int togo = phase == 0 ? 0 : s.div - phase;
    // (the tick loop holds one s_waitcnt vmcnt(0) although it has no memory instruction — a load of the period's beginning is still pending on one path into it.
    //  Waiting in front of the loop instead removes 22 scalar instructions per period and is 0.3 % SLOWER, 17.19 against 17.14 us: the first tick hides that
    //  load's latency; profiles/r5_ab_tick_loop.txt)
    // (round 5 tried the runs of plain ticks between two manager ticks as an inner loop of their own — one counter, one compare + branch per tick, the tick's body
    //  twice in the code: 17.47 against 17.18 us per period, profiles/r5_ab_tick_loop.txt)
#pragma unroll TickUnroll<T>::n
    for (int i = 0; i < n_ticks; ++i) {
#ifndef DQL_AB_NO_F64_LDS_CONSTS
      if constexpr (sizeof(T) == 8) asm volatile("" ::: "memory");  // float64: the tick's constants live in LDS (k_step) and are read again in every tick, not held in registers
#endif
      DQL_SECTION("rot");
      quat_to_R(e.q, R); yaw_cs(R, cy, sy, ct, rn, c375);
      if (togo == 0) { manager_tick(); togo = s.div; }
      --togo;
      control_and_plant();
    }
  } else {
    // one wave per SIMD (small batches, registers are free): the 21 / 22 ticks of a period = a few ticks up to the next manager
    // tick, then whole manager periods (one manager tick + DQL_GROUP physics ticks, straight-line: no phase test, the filter
    // histories rotate by renaming instead of moves), then the rest.  Same operations in the same order as the plain loop,
    // which still serves any other manager_div.  Measured: -4 % at 4 096 envs.
    int left = n_ticks;
    for (;;) {
      while (left > 0 && !(s.div == DQL_GROUP && phase == 0 && left >= DQL_GROUP)) {
        DQL_SECTION("rot");
        quat_to_R(e.q, R); yaw_cs(R, cy, sy, ct, rn);
        if (phase == 0) manager_tick();
        phase = (phase + 1 == s.div) ? 0 : phase + 1;
        control_and_plant();
        --left;
      }
      if (left == 0) break;
      do {
        DQL_SECTION("rot");
        quat_to_R(e.q, R); yaw_cs(R, cy, sy, ct, rn);
        manager_tick();
        control_and_plant();
#pragma unroll
        for (int k = 1; k < DQL_GROUP; ++k) {
          quat_to_R(e.q, R); yaw_cs(R, cy, sy, ct, rn);
          control_and_plant();
        }
        left -= DQL_GROUP;
      } while (left >= DQL_GROUP);
    }
  }
  DQL_SECTION("epilogue");
  DQL_MARK_T(e, 4);
  DQL_PHASE(e, 2);
#if defined(DQL_SCALAR_MDP_ALL)   // A/B builds (tools/ab_build.sh)
  const StepOut o = period_end<MDP_SCALAR>(s, mp, mr, e, c, qa, qb, mode);
#elif defined(DQL_SCALAR_MDP_NONE)
  const StepOut o = period_end<MDP_VECTOR>(s, mp, mr, e, c, qa, qb, mode);
#elif defined(DQL_AB_NO_LITERAL_MDP)
  const StepOut o = period_end<HOT ? MDP_VECTOR : MDP_SCALAR>(s, mp, mr, e, c, qa, qb, mode);
#else
  const StepOut o = period_end<(TICK == TICK_LIT || TICK == TICK_PACKED_LITM) ? MDP_LITERAL : (HOT ? MDP_VECTOR : MDP_SCALAR)>(s, mp, mr, e, c, qa, qb, mode);
#endif
  DQL_MARK_T(e, 5);
  DQL_PHASE(e, 4);
  return o;
}

// ---------------------------------------------------------------------------------------------
// HBM layout: real fields as quads [NQ_REAL][n_pad] of Quad<T>; int fields as int4 [n_pad]
//   int4 = { idx_x, idx_y, step_count | cur_check << 16, code | flags << 8 | action << 16 }
// x-axis configs touch quads 0-10 (+13 when the platform is per-env) and write quad 14.
// ---------------------------------------------------------------------------------------------
template <typename T> DQL_DEV void load_env(Env<T>& e, const Quad<T>* __restrict__ sr, const int4 iv, long long n, long long i, const SimK<T>& c) {
  const Quad<T> q0 = sr[0 * n + i], q1 = sr[1 * n + i], q2 = sr[2 * n + i], q3 = sr[3 * n + i], q4 = sr[4 * n + i], q5 = sr[5 * n + i];
  const Quad<T> q6 = sr[6 * n + i], q7 = sr[7 * n + i], q8 = sr[8 * n + i], q9 = sr[9 * n + i], q10 = sr[10 * n + i];
  e.p[0] = q0.a; e.p[1] = q0.b; e.p[2] = q0.c; e.v[0] = q0.d;
  e.v[1] = q1.a; e.v[2] = q1.b; e.q[0] = q1.c; e.q[1] = q1.d;
  e.q[2] = q2.a; e.q[3] = q2.b; e.w[0] = q2.c; e.w[1] = q2.d;
  e.w[2] = q3.a; e.om[0] = q3.b; e.om[1] = q3.c; e.om[2] = q3.d;
  e.om[3] = q4.a; e.vz_i = q4.b; e.vz_x1 = q4.c; e.vz_x2 = q4.d;
  e.vz_y1 = q5.a; e.vz_y2 = q5.b; e.vz_y3 = q5.c; e.vz_state = q5.d;
  e.yw_i = q6.a; e.yw_x1 = q6.b; e.yw_x2 = q6.c; e.yw_y1 = q6.d;
  e.yw_y2 = q7.a; e.yw_y3 = q7.b; e.yw_state = q7.c; e.pitch_sp = q7.d;
  e.mp_phase = q8.a; e.mp_x = q8.b; e.mp_u = q8.c; e.vf_x = q8.d;
  e.kal_x_x = q9.a; e.kal_x_P = q9.b; e.shp_p = q9.c; e.shp_v = q9.d;
  e.shp_a = q10.a; e.cum_x = q10.b; e.roll_sp = q10.c; e.mp_y = q10.d;
  if (c.two_axis || c.traj == DQL_TRAJ_EIGHT) {
    const Quad<T> q11 = sr[11 * n + i];
    e.mp_v = q11.a; e.vf_y = q11.b; e.kal_y_x = q11.c; e.kal_y_P = q11.d;
  } else { e.mp_v = T(0.0); e.vf_y = T(0.0); e.kal_y_x = T(0.0); e.kal_y_P = T(1.0); }
  if (c.two_axis) { const Quad<T> q12 = sr[12 * n + i]; e.shpy_p = q12.a; e.shpy_v = q12.b; e.shpy_a = q12.c; e.cum_y = q12.d; }
  else { e.shpy_p = e.shpy_v = e.shpy_a = e.cum_y = T(0.0); }
  if (c.per_env_platform) { const Quad<T> q13 = sr[13 * n + i]; e.mp_r = q13.a; e.mp_w = q13.b; }
  else { e.mp_r = c.mp_r; e.mp_w = c.mp_w; }
  e.idx_x = iv.x; e.idx_y = iv.y; e.step_count = iv.z & 0xffff; e.cur_check = (iv.z >> 16) & 0xffff;
  e.bin_k = idx_level(iv.x); e.bin_p = idx_pos(iv.x); e.bin_ky = idx_level(iv.y); e.bin_py = idx_pos(iv.y);
  e.code = iv.w & 0xff; e.flags = (iv.w >> 8) & 0xff; e.action = (iv.w >> 16) & 0xff;
  e.reward = T(0.0); e.obs_px = e.obs_vx = e.obs_ax = e.obs_py = e.obs_vy = e.obs_ay = T(0.0);
}
template <typename T> DQL_DEV void store_env(const Env<T>& e, Quad<T>* __restrict__ sr, int4* __restrict__ si, long long n, long long i,
                                             const SimK<T>& c) {
  sr[0 * n + i] = Quad<T>{opq(e.p[0]), opq(e.p[1]), opq(e.p[2]), opq(e.v[0])};
  sr[1 * n + i] = Quad<T>{opq(e.v[1]), opq(e.v[2]), opq(e.q[0]), opq(e.q[1])};
  sr[2 * n + i] = Quad<T>{opq(e.q[2]), opq(e.q[3]), opq(e.w[0]), opq(e.w[1])};
  sr[3 * n + i] = Quad<T>{opq(e.w[2]), opq(e.om[0]), opq(e.om[1]), opq(e.om[2])};
  sr[4 * n + i] = Quad<T>{opq(e.om[3]), opq(e.vz_i), opq(e.vz_x1), opq(e.vz_x2)};
  sr[5 * n + i] = Quad<T>{opq(e.vz_y1), opq(e.vz_y2), opq(e.vz_y3), opq(e.vz_state)};
  sr[6 * n + i] = Quad<T>{opq(e.yw_i), opq(e.yw_x1), opq(e.yw_x2), opq(e.yw_y1)};
  sr[7 * n + i] = Quad<T>{opq(e.yw_y2), opq(e.yw_y3), opq(e.yw_state), opq(e.pitch_sp)};
  sr[8 * n + i] = Quad<T>{opq(e.mp_phase), opq(e.mp_x), opq(e.mp_u), opq(e.vf_x)};
  sr[9 * n + i] = Quad<T>{opq(e.kal_x_x), opq(e.kal_x_P), opq(e.shp_p), opq(e.shp_v)};
  sr[10 * n + i] = Quad<T>{opq(e.shp_a), opq(e.cum_x), opq(e.roll_sp), opq(e.mp_y)};
  if (c.two_axis || c.traj == DQL_TRAJ_EIGHT) sr[11 * n + i] = Quad<T>{opq(e.mp_v), opq(e.vf_y), opq(e.kal_y_x), opq(e.kal_y_P)};
  if (c.two_axis) sr[12 * n + i] = Quad<T>{opq(e.shpy_p), opq(e.shpy_v), opq(e.shpy_a), opq(e.cum_y)};
  sr[14 * n + i] = Quad<T>{opq(e.reward), opq(e.obs_px), opq(e.obs_vx), opq(e.obs_ax)};
  sr[15 * n + i] = Quad<T>{opq(e.obs_py), opq(e.obs_vy), opq(e.obs_ay), T(0.0)};
  si[i] = make_int4(e.idx_x, e.idx_y, (e.step_count & 0xffff) | (e.cur_check << 16), (e.code & 0xff) | ((e.flags & 0xff) << 8) | ((e.action & 0xff) << 16));
}

}  // namespace dql
