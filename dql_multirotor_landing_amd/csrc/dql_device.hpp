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
  const T t1 = w * (T(3.999999999940941908e-01) + w * (T(2.222219843214978396e-01) + w * T(1.531383769920937332e-01)));
  const T t2 = z * (T(6.666666666666735130e-01) + w * (T(2.857142874366239149e-01) + w * (T(1.818357216161805012e-01) +
               w * T(1.479819860511658591e-01))));
  const T Rr = t2 + t1, hfsq = T(0.5) * f * f, dk = (T)k;
  return dk * T(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + Rr) + dk * T(1.90821492927058770002e-10))) - f);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------------------------
DQL_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
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
template <typename T> struct MdpK {
  T p_max, v_max, a_max, theta_max, delta_theta, beta, sigma_a, min_alt;
  T w_p, w_v, w_theta, w_dur, w_fail, w_succ, delta_t, f_ag, timeout_steps;
  T lim_p[5], lim_v[5], lim_a[5], angles[7];
  double gamma;
  int working, goal_logic;
  uint32_t quirks;
};
// SimK: constants of the physics tick loop (kernel argument by value).
template <typename T> struct SimK {
  T dt, g, inv_m, I[3], inv_I[3], l, h, kf, km, lkf, kmkf, aup, adn, omax, cd, crd;
  T kR[3], kW[3], ia, ib, ic;
  T vz_kp, vz_ki, vz_lo, vz_hi, vz_wind, vz_sp;
  T yw_kp, yw_ki, yw_lo, yw_hi, yw_wind, yw_sp;
  T bw_k1, bw_k2, bw_inv;
  T mp_dt, mp_top, mp_hx, mp_hy, bottom;
  T noise_p, noise_v, kal_q, kal_r, mgr_dt, mp_r, mp_w;
  T p_max, theta_max, delta_theta, z_init, init_sigma;  // reset placement / set-point update (before the loop)
  int div, traj, init_uniform, working, per_env_platform, two_axis;
  uint32_t quirks;
};

// Per-tick constants held in VECTOR registers for the duration of the tick loop.  The loop wants ~50 constants on top of its
// loop state; as wave-uniform scalars they overflow the SGPR file and every use of a spilled one costs a v_readlane.  A VALU
// operand may just as well be a VGPR: one v_mov per constant before the loop (opaque to the compiler, so it stays there).
DQL_DEV float to_vgpr(float x) { float y; asm("v_mov_b32 %0, %1" : "=v"(y) : "s"(x)); return y; }
template <typename T> struct HotK {
  T dt, g, inv_m, I[3], inv_I[3], l, h, kf, lkf, kmkf, aup, adn, omax, cd, crd;
  T kR[3], kW[3], ia, ib, ic;
  T vz_kp, vz_ki, vz_lo, vz_hi, vz_wind, vz_sp;
  T yw_kp, yw_ki, yw_lo, yw_hi, yw_wind, yw_sp;
  T bw_k1, bw_k2, bw_inv;
  T mp_top, mp_hx, mp_hy, bottom;
};
DQL_DEV HotK<float> make_hot(const SimK<float>& s) {
  HotK<float> h;
#define DQL_HOT(f) h.f = to_vgpr(s.f)
  DQL_HOT(dt); DQL_HOT(g); DQL_HOT(inv_m); DQL_HOT(I[0]); DQL_HOT(I[1]); DQL_HOT(I[2]); DQL_HOT(inv_I[0]); DQL_HOT(inv_I[1]); DQL_HOT(inv_I[2]);
  DQL_HOT(l); DQL_HOT(h); DQL_HOT(kf); DQL_HOT(lkf); DQL_HOT(kmkf); DQL_HOT(aup); DQL_HOT(adn); DQL_HOT(omax); DQL_HOT(cd); DQL_HOT(crd);
  DQL_HOT(kR[0]); DQL_HOT(kR[1]); DQL_HOT(kR[2]); DQL_HOT(kW[0]); DQL_HOT(kW[1]); DQL_HOT(kW[2]); DQL_HOT(ia); DQL_HOT(ib); DQL_HOT(ic);
  DQL_HOT(vz_kp); DQL_HOT(vz_ki); DQL_HOT(vz_lo); DQL_HOT(vz_hi); DQL_HOT(vz_wind); DQL_HOT(vz_sp);
  DQL_HOT(yw_kp); DQL_HOT(yw_ki); DQL_HOT(yw_lo); DQL_HOT(yw_hi); DQL_HOT(yw_wind); DQL_HOT(yw_sp);
  DQL_HOT(bw_k1); DQL_HOT(bw_inv); DQL_HOT(mp_top); DQL_HOT(mp_hx); DQL_HOT(mp_hy); DQL_HOT(bottom);
#undef DQL_HOT
  h.bw_k2 = s.bw_k2;  // only steers a wave-uniform branch
  return h;
}

enum { FL_DONE = 1, FL_CONTACT = 2, FL_ACC_INIT = 4, FL_WAS_RESET = 8, FL_OBS_CONTACT = 16 };
enum { MODE_TRAIN = 0, MODE_EVAL = 1, MODE_EXTERNAL = 2 };

constexpr int NQ_REAL = 16;  // quads of real fields per env (64 fields)
constexpr int NF_REAL = 64, NF_INT = 7;

// per-env state in registers (field order = quad layout, see dql_field_name)
template <typename T> struct Env {
#ifdef DQL_WAVE_CLOCK
  unsigned long long mark;
#endif
  T p[3], v[3], q[4], w[3], om[4];
  T vz_i, vz_x1, vz_x2, vz_y1, vz_y2, vz_y3, vz_state;
  T yw_i, yw_x1, yw_x2, yw_y1, yw_y2, yw_y3, yw_state;
  T pitch_sp, mp_phase, mp_x, mp_u, vf_x, kal_x_x, kal_x_P, shp_p, shp_v, shp_a, cum_x, roll_sp, mp_y;
  T mp_v, vf_y, kal_y_x, kal_y_P, mp_r, mp_w;
  T shpy_p, shpy_v, shpy_a, cum_y;
  T reward, obs_px, obs_vx, obs_ax, obs_py, obs_vy, obs_ay;
  int idx_x, idx_y, step_count, cur_check, code, flags, action;
};

template <typename T> DQL_DEV T sel5(const T (&a)[5], int k) {
  return k == 0 ? a[0] : k == 1 ? a[1] : k == 2 ? a[2] : k == 3 ? a[3] : a[4];
}

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
template <typename T> DQL_DEV int disc3(T v, T goal, T limit) {  // :160-170
  if (-limit <= v && v < -goal) return 0;
  if (-goal <= v && v <= goal) return 1;
  if (v <= limit) return 2;
  return -1;
}
template <typename T> DQL_DEV int discretise(const MdpK<T>& m, T rel_p, T rel_v, T rel_a, T angle) {  // :257-333
  const T cp = clip(rel_p / m.p_max, T(-1.0), T(1.0));
  const T cv = clip(rel_v / m.v_max, T(-1.0), T(1.0));
  const T ca = clip(rel_a / m.a_max, T(-1.0), T(1.0));
  const int n = m.working + 1;
  int k = latest_valid_level(m.lim_p, n, cp);
  const int kv = latest_valid_level(m.lim_v, n, cv), ka = latest_valid_level(m.lim_a, n, ca);
  k = kv < k ? kv : k;
  k = ka < k ? ka : k;
  const T lp = sel5(m.lim_p, k), lv = sel5(m.lim_v, k), la = sel5(m.lim_a, k);
  T pc = m.beta, vc = m.beta, ac = m.sigma_a;
  if (k < m.working) { pc = sel5(m.lim_p, k + 1) / lp; vc = sel5(m.lim_v, k + 1) / lv; }
  if (k == m.working) ac = ac * m.beta;
  const int dp = disc3(cp, lp * pc, lp);
  const int dv = disc3(cv, lv * vc, lv);
  const int da = disc3(ca, la * ac, la);
  if (dp < 0 || dv < 0 || da < 0) return -1;
  const T ct = clip(angle, -m.theta_max, m.theta_max);
  int best = 0; T bd = abs_(m.angles[0] - ct);
#pragma unroll
  for (int i = 1; i < 7; ++i) { const T d = abs_(m.angles[i] - ct); if (d < bd) { bd = d; best = i; } }
  return (((k * 3 + dp) * 3 + dv) * 3 + da) * 7 + best;
}
DQL_DEV int idx_level(int idx) { return idx / DQL_STATES_PER_LEVEL; }
DQL_DEV int idx_pos(int idx) { return (idx / 63) % 3; }
DQL_DEV int idx_vel(int idx) { return (idx / 21) % 3; }

template <typename T, typename K> DQL_DEV T continuous_action(const K& m, T sp, int action) {  // :543-560
  if (action == 0) { const T t = sp + m.delta_theta; return t < m.theta_max ? t : m.theta_max; }
  if (action == 1) { const T t = sp - m.delta_theta; return t > -m.theta_max ? t : -m.theta_max; }
  return sp;
}
template <typename T>
DQL_DEV int mdp_check(const MdpK<T>& m, int& step_count, int& cur_check, int code, int prev_idx, int cur_idx, bool contact, T rel_p_x,
                      T rel_p_y, T abs_p_z, bool two = false, int prev_idy = -1, int cur_idy = -1) {  // :335-439
  // two-axis configs (beyond the reference, B16): the goal state is the joint goal of both 1-D MDPs
  const bool goal_x = prev_idx >= 0 && idx_pos(cur_idx) == 1 && idx_vel(cur_idx) == 1;
  const bool goal_y = !two || (prev_idy >= 0 && idx_pos(cur_idy) == 1 && idx_vel(cur_idy) == 1);
  const bool lvl_x = idx_level(prev_idx) == m.working && idx_level(cur_idx) == m.working;
  const bool lvl_y = !two || (idx_level(prev_idy) == m.working && idx_level(cur_idy) == m.working);
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
template <typename T>
DQL_DEV T mdp_reward(const MdpK<T>& m, T& shp_p, T& shp_v, T& shp_a, T& cum, int code, int cur_idx, T rel_p, T rel_v, T angle_sp) {  // :441-541
  const T ncp = clip(rel_p / m.p_max, T(-1.0), T(1.0));
  const T ncv = clip(rel_v / m.v_max, T(-1.0), T(1.0));
  const T npitch = angle_sp / m.theta_max;
  const int k = idx_level(cur_idx);
  const T lv = sel5(m.lim_v, k), la = sel5(m.lim_a, k);
  const T prev_p = shp_p, prev_v = shp_v, prev_a = shp_a;
  shp_p = m.w_p * abs_(ncp); shp_v = m.w_v * abs_(ncv); shp_a = m.w_theta * abs_(npitch);
  const T r_p_max = abs_(m.w_p) * lv * m.delta_t;
  const T r_v_max = abs_(m.w_v) * la * m.delta_t;
  const T r_theta_max = abs_(m.w_theta) * (m.delta_theta / m.theta_max) * lv;
  const T r_dur_max = m.w_dur * lv * m.delta_t;
  const T r_max = r_p_max + r_v_max + r_theta_max + r_dur_max;
  const T r_p = clip(shp_p - prev_p, -r_p_max, r_p_max);
  const T r_v = clip(shp_v - prev_v, -r_v_max, r_v_max);
  const T r_theta = m.w_theta * (abs_(shp_a) - abs_(prev_a)) / m.theta_max * lv;
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
template <typename T, typename K> DQL_DEV T butterworth(const K& c, T x0, T& x1, T& x2, T& y1, T& y2, T& y3) {  // filters.py:98-109
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
  integ = clip3(integ + e0 * c.dt, -wind, wind);
  const T fe = butterworth(c, e0, x1, x2, y1, y2, y3);
  return clip3(kp * fe + ki * integ, lo, hi);
}
template <typename T> DQL_DEV T kalman1d(T& x, T& P, T Q, T Rm, T z) {  // filters.py:19-36
  P += Q;
  if (Rm == T(0.0)) {  // wave-uniform (launch files: no measurement noise): P / (P + 0) is exactly 1, no division to pay for
    x += (z - x);
    P *= T(0.0);
    return x;
  }
  const T K = P / (P + Rm);
  x += K * (z - x);
  P *= (T(1.0) - K);
  return x;
}

// ---------------------------------------------------------------------------------------------
// simulator pieces
// ---------------------------------------------------------------------------------------------
template <typename T> DQL_DEV void quat_to_R(const T (&q)[4], T (&R)[9]) {
  const T w = q[0], x = q[1], y = q[2], z = q[3];
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
template <typename T> DQL_DEV void yaw_cs(const T (&R)[9], T& c, T& s) {
  const T n2 = fma_(R[0], R[0], R[3] * R[3]);
  const T h = T(-0.5) * n2;
  T r = fma_(T(-0.5), n2, T(1.5));
#pragma unroll
  for (int k = 0; k < YawIters<T>::n; ++k) r = r * fma_(h * r, r, T(1.5));
  c = R[0] * r; s = R[3] * r;
}
// attitude_controller.py:107-156
template <typename T, typename K>
DQL_DEV void attitude(const K& s, const T (&R)[9], const T (&w)[3], const T (&B)[9], T cy, T sy, T r_cmd, T thrust, T (&cmd)[4]) {
  T D[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) { D[j] = fma_(cy, B[j], -(sy * B[3 + j])); D[3 + j] = fma_(sy, B[j], cy * B[3 + j]); D[6 + j] = B[6 + j]; }
#define DQL_E(i, j) fma_(D[i], R[j], fma_(D[3 + i], R[3 + j], D[6 + i] * R[6 + j]))
  const T E01 = DQL_E(0, 1), E10 = DQL_E(1, 0), E02 = DQL_E(0, 2), E20 = DQL_E(2, 0), E12 = DQL_E(1, 2), E21 = DQL_E(2, 1), E22 = DQL_E(2, 2);
#undef DQL_E
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
// gazebo_motor_model.cpp:434-500 + semi-implicit Euler of one rigid body
// first-order rotor speed filter (common.h:147-183), commanded speed clipped at max_rot_velocity (gazebo_motor_model.cpp:358-364)
template <typename T, typename K> DQL_DEV void rotor_filter(const K& s, Env<T>& e, const T (&cmd)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T ref = clip3(cmd[i], T(0.0), T(s.omax));  // cmd = sqrt(..) >= +0: min(cmd, omax)
    const T a = ref > e.om[i] ? s.aup : s.adn;
    e.om[i] = fma_(a, e.om[i], (T(1.0) - a) * ref);
  }
}
// forces from the CURRENT rotor speeds (gazebo_motor_model.cpp:434-500) + semi-implicit Euler of one rigid body
template <typename T, typename K> DQL_DEV void plant_step(const K& s, Env<T>& e, const T (&R)[9]) {
  const T l = s.l, h = s.h;
  const T w0 = e.w[0], w1 = e.w[1], w2 = e.w[2];
  // thrust k_f om_i^2 along body z at rotor i = (+l,0,h), (0,+l,h), (-l,0,h), (0,-l,h); drag torque -dir_i k_m T_i
  const T q0 = e.om[0] * e.om[0], q1 = e.om[1] * e.om[1], q2 = e.om[2] * e.om[2], q3 = e.om[3] * e.om[3];
  const T Fbz = s.kf * ((q0 + q1) + (q2 + q3));
  T tx = s.lkf * (q1 - q3), ty = s.lkf * (q2 - q0), tz = s.kmkf * ((q0 - q1) + (q2 - q3));
  // rotor drag -|om_i| c_d v_perp,i with v_perp,i = (v_body + w x r_i) restricted to the rotor plane, summed in closed form:
  // sum_i om_i (w x r_i)_x = S w_y h - w_z l (om1 - om3),  sum_i om_i (w x r_i)_y = -S w_x h + w_z l (om0 - om2)
  const T vbx = fma_(R[0], e.v[0], fma_(R[3], e.v[1], R[6] * e.v[2]));
  const T vby = fma_(R[1], e.v[0], fma_(R[4], e.v[1], R[7] * e.v[2]));
  const T uxc = fma_(w1, h, vbx), uyc = fma_(-w0, h, vby), wzl = w2 * l;
  const T S = (e.om[0] + e.om[1]) + (e.om[2] + e.om[3]), d02 = e.om[0] - e.om[2], d13 = e.om[1] - e.om[3];
  const T Fbx = -(s.cd * fma_(S, uxc, -(wzl * d13)));
  const T Fby = -(s.cd * fma_(S, uyc, wzl * d02));
  const T tzd = -(s.cd * fma_(uyc, d02, fma_(wzl, S, -(uxc * d13))));  // sum_i (r_i x drag_i)_z / l
  tx = fma_(-h, Fby, tx); ty = fma_(h, Fbx, ty); tz = fma_(l, tzd, tz);
  tx = fma_(s.crd, Fbx, tx); ty = fma_(s.crd, Fby, ty);  // rolling moment = (c_r / c_d) * drag force
  const T ax = fma_(R[0], Fbx, fma_(R[1], Fby, R[2] * Fbz)) * s.inv_m;
  const T ay = fma_(R[3], Fbx, fma_(R[4], Fby, R[5] * Fbz)) * s.inv_m;
  const T az = fma_(R[6], Fbx, fma_(R[7], Fby, R[8] * Fbz)) * s.inv_m - s.g;
  e.v[0] = fma_(s.dt, ax, e.v[0]); e.v[1] = fma_(s.dt, ay, e.v[1]); e.v[2] = fma_(s.dt, az, e.v[2]);
  e.p[0] = fma_(s.dt, e.v[0], e.p[0]); e.p[1] = fma_(s.dt, e.v[1], e.p[1]); e.p[2] = fma_(s.dt, e.v[2], e.p[2]);
  const T Iw0 = s.I[0] * w0, Iw1 = s.I[1] * w1, Iw2 = s.I[2] * w2;
  const T g0 = fma_(w1, Iw2, -(w2 * Iw1)), g1 = fma_(w2, Iw0, -(w0 * Iw2)), g2 = fma_(w0, Iw1, -(w1 * Iw0));
  e.w[0] = fma_(s.dt, (tx - g0) * s.inv_I[0], w0);
  e.w[1] = fma_(s.dt, (ty - g1) * s.inv_I[1], w1);
  e.w[2] = fma_(s.dt, (tz - g2) * s.inv_I[2], w2);
  const T qw = e.q[0], qx = e.q[1], qy = e.q[2], qz = e.q[3], hdt = T(0.5) * s.dt;
  const T dw = -fma_(qx, e.w[0], fma_(qy, e.w[1], qz * e.w[2]));
  const T dxq = fma_(qw, e.w[0], fma_(qy, e.w[2], -(qz * e.w[1])));
  const T dyq = fma_(qw, e.w[1], fma_(qz, e.w[0], -(qx * e.w[2])));
  const T dzq = fma_(qw, e.w[2], fma_(qx, e.w[1], -(qy * e.w[0])));
  const T nw = fma_(hdt, dw, qw), nx = fma_(hdt, dxq, qx), ny = fma_(hdt, dyq, qy), nz = fma_(hdt, dzq, qz);
  // renormalise with one Newton step of 1/sqrt(|q|^2) about 1: |q|^2 - 1 = O((dt |w|)^2), so the residual is O(dt^4)
  const T inv = fma_(T(-0.5), fma_(nw, nw, fma_(nx, nx, fma_(ny, ny, nz * nz))), T(1.5));
  e.q[0] = nw * inv; e.q[1] = nx * inv; e.q[2] = ny * inv; e.q[3] = nz * inv;
}
// moving_platform.py:87-127
template <typename T> DQL_DEV void platform_eval(const SimK<T>& s, Env<T>& e) {
  T sn, cs;
  det_sincos(e.mp_phase, sn, cs);
  if (s.traj == DQL_TRAJ_EIGHT) {
    e.mp_x = e.mp_r * cs; e.mp_y = e.mp_r * sn * cs;
    e.mp_u = -(e.mp_r * e.mp_w) * sn; e.mp_v = e.mp_r * e.mp_w * (cs * cs - sn * sn);
  } else {
    e.mp_x = e.mp_r * sn; e.mp_y = T(0.0);
    e.mp_u = e.mp_r * e.mp_w * cs; e.mp_v = T(0.0);
  }
}
template <typename T> DQL_DEV void platform_update(const SimK<T>& s, Env<T>& e) {
  platform_eval(s, e);
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
template <typename T>
DQL_DEV void manager_obs(const SimK<T>& s, Env<T>& e, T cy, T sy, long long mgr_index, uint32_t k0, uint32_t k1,
                         uint32_t step_lo, uint32_t step_hi, uint32_t env_id, uint32_t mgr_in_step) {
  const T dxw = e.mp_x - e.p[0], dyw = e.mp_y - e.p[1];
  const T dvx = e.mp_u - e.v[0], dvy = e.mp_v - e.v[1];
  const T rpx = fma_(cy, dxw, sy * dyw), rpy = fma_(cy, dyw, -(sy * dxw));
  const T rvx = fma_(cy, dvx, sy * dvy), rvy = fma_(cy, dvy, -(sy * dvx));
  T opx = rpx, opy = rpy, ovx = rvx, ovy = rvy;
  if (s.noise_p > T(0.0) || s.noise_v > T(0.0)) {
    uint32_t r[4]; T n0, n1, n2, n3;
    philox4x32(step_lo, step_hi, env_id, STREAM_NOISE0 + mgr_in_step, k0, k1, r);
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
    ax_ = kalman1d(e.kal_x_x, e.kal_x_P, s.kal_q, s.kal_r, (rvx - e.vf_x) / dt_);
    if (s.two_axis) ay_ = kalman1d(e.kal_y_x, e.kal_y_P, s.kal_q, s.kal_r, (rvy - e.vf_y) / dt_);
    if (!(s.quirks & DQL_Q_FROZEN_ACC_REFERENCE)) { e.vf_x = rvx; if (s.two_axis) e.vf_y = rvy; }
  }
  e.obs_px = opx; e.obs_py = opy; e.obs_vx = ovx; e.obs_vy = ovy; e.obs_ax = ax_; e.obs_ay = ay_;
  if (e.flags & FL_CONTACT) e.flags |= FL_OBS_CONTACT; else e.flags &= ~FL_OBS_CONTACT;
  platform_update(s, e);
}
// platform extrapolation between manager ticks + bumper contact test
template <typename T, typename K> DQL_DEV void platform_contact(const K& s, Env<T>& e) {
  e.mp_x = fma_(e.mp_u, s.dt, e.mp_x); e.mp_y = fma_(e.mp_v, s.dt, e.mp_y);
  if (e.p[2] - s.bottom <= s.mp_top && abs_(e.p[0] - e.mp_x) <= s.mp_hx && abs_(e.p[1] - e.mp_y) <= s.mp_hy) e.flags |= FL_CONTACT;
}
// B = Rx(roll_sp) Ry(pitch_sp) (attitude_controller.py:138-140), constant over one agent period
template <typename T> DQL_DEV void make_B(T pitch_sp, T roll_sp, T (&B)[9]) {
  T sp_, cp_, sr_, cr_;
  det_sincos(pitch_sp, sp_, cp_); det_sincos(roll_sp, sr_, cr_);
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

struct StepOut {  // what one env contributes to the shared tables / counters this period
  long long target_fx;  // TD target, fixed point (DQL_TARGET_FRAC_BITS)
  long long target_y_fx;
  long long reward_fx;
  int cell, cell_y;     // table * N_CELLS + idx*3+action (table 0 = Q_table_a, 1 = Q_table_b), or -1
  int decision, done;
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
DQL_DEV PeriodCtx period_begin(const SimK<T>& s, Env<T>& e, const QRow& qx, TabPtr qa, TabPtr qb, int mode, double eps, int ext_action, uint64_t seed,
                               uint32_t env_id, long long step_index) {
  PeriodCtx c;
  c.k0 = (uint32_t)seed; c.k1 = (uint32_t)(seed >> 32); c.step_lo = (uint32_t)step_index; c.step_hi = (uint32_t)((uint64_t)step_index >> 32);
  uint32_t r[4];
  philox4x32(c.step_lo, c.step_hi, env_id, STREAM_ACTION, c.k0, c.k1, r);
  c.is_reset = (e.flags & FL_DONE) != 0;
  c.prev_idx = e.idx_x; c.prev_idy = e.idx_y;
  const bool two = s.two_axis != 0;
  int action = 2, action_y = 2;
  uint32_t r2[4] = {0u, 0u, 0u, 0u};
  if (two) philox4x32(c.step_lo, c.step_hi, env_id, STREAM_ACTION + 1u, c.k0, c.k1, r2);
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
      const bool explore = (mode == MODE_TRAIN) && ((double)u24<T>(r[0]) < eps);
      action = explore ? (int)(((uint64_t)r[1] * 3u) >> 32) : greedy;
      if (two) {
        const int greedy_y = agent_predict(qa, qb, c.prev_idy < 0 ? 0 : c.prev_idy);
        const bool explore_y = (mode == MODE_TRAIN) && ((double)u24<T>(r2[0]) < eps);
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
template <typename T, typename TabPtr>
DQL_DEV StepOut period_end(const SimK<T>& s, const MdpK<T>* __restrict__ mp, Env<T>& e, const PeriodCtx& c, TabPtr qa, TabPtr qb, int mode) {
  StepOut out; out.cell = -1; out.cell_y = -1; out.decision = 0; out.done = 0; out.target_fx = 0; out.target_y_fx = 0; out.reward_fx = 0;
  const bool two = s.two_axis != 0;
  const int prev_idx = c.prev_idx, prev_idy = c.prev_idy;
  asm volatile("" ::: "memory");  // keep the MdpK scalar loads below the tick loop
  const MdpK<T> m = *mp;
  T R[9];
  quat_to_R(e.q, R);
  const T cyy = sqrt_(fma_(R[0], R[0], R[3] * R[3]));
  const T pitch = det_atan2(-R[6], cyy);
  int idx = discretise(m, e.obs_px, e.obs_vx, e.obs_ax, pitch);
  if (idx < 0) idx = 0;
  e.idx_x = idx;
  int idy = -1;
  if (two) {
    const T roll = det_atan2(R[7], R[8]);
    idy = discretise(m, e.obs_py, e.obs_vy, e.obs_ay, -roll);
    if (idy < 0) idy = 0;
    e.idx_y = idy;
  }
  e.reward = T(0.0);
  if (c.is_reset) return out;
  const bool contact = (e.flags & FL_OBS_CONTACT) != 0;
  e.code = mdp_check(m, e.step_count, e.cur_check, e.code, prev_idx, idx, contact, e.obs_px, e.obs_py, e.p[2], two, prev_idy, idy);
  const T rew = mdp_reward(m, e.shp_p, e.shp_v, e.shp_a, e.cum_x, e.code, idx, e.obs_px, e.obs_vx, e.pitch_sp);
  T rew_y = T(0.0);
  if (two) rew_y = mdp_reward(m, e.shpy_p, e.shpy_v, e.shpy_a, e.cum_y, e.code, idy, e.obs_py, e.obs_vy, -e.roll_sp);
  e.reward = two ? rew + rew_y : rew;
  const bool done = e.code <= DQL_TERMINAL_TIMEOUT;
  if (done) e.flags |= FL_DONE;
  out.decision = 1; out.done = done ? 1 : 0;
  out.reward_fx = __double2ll_rn((double)rew * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (two) out.reward_fx += __double2ll_rn((double)rew_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (mode == MODE_TRAIN) {
    // Reference (B1/B2, DQL_Q_UPDATE_TABLE_A_ONLY): always Q_table_a, valued by itself.  Otherwise Double Q-learning as the paper
    // has it: the coin picks the table to update, the OTHER table values the picked table's greedy action at s'.
    const bool dbl = !(s.quirks & DQL_Q_UPDATE_TABLE_A_ONLY);
    {
      const bool sel_b = dbl && c.coin;
      const double a0 = qa[idx * 3], a1 = qa[idx * 3 + 1], a2 = qa[idx * 3 + 2];
      double s0 = a0, s1 = a1, s2 = a2, v0 = a0, v1 = a1, v2 = a2;
      if (dbl) {
        const double b0 = qb[idx * 3], b1 = qb[idx * 3 + 1], b2 = qb[idx * 3 + 2];
        s0 = sel_b ? b0 : a0; s1 = sel_b ? b1 : a1; s2 = sel_b ? b2 : a2;
        v0 = sel_b ? a0 : b0; v1 = sel_b ? a1 : b1; v2 = sel_b ? a2 : b2;
      }
      const int b = argmax3(s0, s1, s2);
      const double boot = b == 0 ? v0 : (b == 1 ? v1 : v2);
      int mask;
      if (s.quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask = idx_pos(prev_idx) != idx_pos(idx);
      else mask = !done;
      const double target = (double)rew + (m.gamma * boot) * (double)mask;
      out.cell = prev_idx * 3 + c.action + (sel_b ? DQL_N_CELLS : 0);
      out.target_fx = __double2ll_rn(target * (double)(1ll << DQL_TARGET_FRAC_BITS));
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
      if (s.quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask_y = idx_pos(prev_idy) != idx_pos(idy);
      else mask_y = !done;
      const double target_y = (double)rew_y + (m.gamma * boot_y) * (double)mask_y;
      out.cell_y = prev_idy * 3 + c.action_y + (sel_b ? DQL_N_CELLS : 0);
      out.target_y_fx = __double2ll_rn(target_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
    }
  }
  return out;
}

// One agent period of one env in one lane.  HOT: hold the per-tick constants in VGPRs (small batches: one wave per SIMD,
// registers are free and every avoided v_readlane shortens the dependency-bound stream; at full occupancy it costs a wave).
template <bool HOT, typename T> struct HotSel { static DQL_DEV const SimK<T>& get(const SimK<T>& s) { return s; } };
template <> struct HotSel<true, float> { static DQL_DEV HotK<float> get(const SimK<float>& s) { return make_hot(s); } };
template <bool HOT, typename T, typename TabPtr>
DQL_DEV StepOut agent_period(const SimK<T>& s, const MdpK<T>* __restrict__ mp, Env<T>& e, const QRow& qx, TabPtr qa, TabPtr qb, int mode, double eps,
                             int ext_action, uint64_t seed, uint32_t env_id, long long step_index, long long g0, int n_ticks) {
  const PeriodCtx c = period_begin(s, e, qx, qa, qb, mode, eps, ext_action, seed, env_id, step_index);
  T B[9];
  make_B(e.pitch_sp, e.roll_sp, B);
  const auto h = HotSel<HOT, T>::get(s);
  DQL_MARK_T(e, 3);
  T R[9], cy, sy;
  uint32_t mgr_in_step = 0;
  int phase = (int)(g0 % s.div);        // physics ticks since the last 100 Hz manager tick (wave-uniform)
  long long mgr_index = g0 / s.div + (phase ? 1 : 0);  // index of the next manager tick
  auto manager_tick = [&]() {
    DQL_SECTION("manager");
    manager_states(R, cy, sy, e.v[2], e.vz_state, e.yw_state);
    manager_obs(s, e, cy, sy, mgr_index, c.k0, c.k1, c.step_lo, c.step_hi, env_id, mgr_in_step);
    ++mgr_in_step; ++mgr_index;
  };
  auto control_and_plant = [&]() {
    DQL_SECTION("pid");
    const T thrust = pid_output(h, h.vz_kp, h.vz_ki, h.vz_lo, h.vz_hi, h.vz_wind, h.vz_sp, e.vz_state, e.vz_i, e.vz_x1, e.vz_x2, e.vz_y1, e.vz_y2, e.vz_y3);
    const T r_cmd = pid_output(h, h.yw_kp, h.yw_ki, h.yw_lo, h.yw_hi, h.yw_wind, h.yw_sp, e.yw_state, e.yw_i, e.yw_x1, e.yw_x2, e.yw_y1, e.yw_y2, e.yw_y3);
    T cmd[4];
    DQL_SECTION("attitude");
    attitude(h, R, e.w, B, cy, sy, r_cmd, thrust, cmd);
    DQL_SECTION("motor_body");
    plant_step(h, e, R);
    rotor_filter(h, e, cmd);
    DQL_SECTION("platform_contact");
    platform_contact(h, e);
  };
  if constexpr (!(HOT && sizeof(T) == 4)) {
    // big batches (several waves per SIMD, registers decide the occupancy): the plain loop
#ifndef DQL_TICK_UNROLL
#define DQL_TICK_UNROLL 2  // measured: -9 % at 1 M envs (fewer loop-carried moves); 3, 4, 6 are worse
#endif
#pragma unroll DQL_TICK_UNROLL
    for (int i = 0; i < n_ticks; ++i) {
      DQL_SECTION("rot");
      quat_to_R(e.q, R); yaw_cs(R, cy, sy);
      if (phase == 0) manager_tick();
      phase = (phase + 1 == s.div) ? 0 : phase + 1;
      control_and_plant();
    }
  } else {
    // one wave per SIMD (small batches, registers are free): the 21 / 22 ticks of a period = a few ticks up to the next manager
    // tick, then whole manager periods (one manager tick + DQL_GROUP physics ticks, straight-line: no phase test, the filter
    // histories rotate by renaming instead of moves), then the rest.  Same operations in the same order as the plain loop,
    // which still serves any other manager_div.  Measured: -4 % at 4 096 envs.
#ifndef DQL_GROUP
#define DQL_GROUP 5  // manager_div of the reference: 500 Hz physics / 100 Hz observation
#endif
    int left = n_ticks;
    for (;;) {
      while (left > 0 && !(s.div == DQL_GROUP && phase == 0 && left >= DQL_GROUP)) {
        DQL_SECTION("rot");
        quat_to_R(e.q, R); yaw_cs(R, cy, sy);
        if (phase == 0) manager_tick();
        phase = (phase + 1 == s.div) ? 0 : phase + 1;
        control_and_plant();
        --left;
      }
      if (left == 0) break;
      do {
        DQL_SECTION("rot");
        quat_to_R(e.q, R); yaw_cs(R, cy, sy);
        manager_tick();
        control_and_plant();
#pragma unroll
        for (int k = 1; k < DQL_GROUP; ++k) {
          quat_to_R(e.q, R); yaw_cs(R, cy, sy);
          control_and_plant();
        }
        left -= DQL_GROUP;
      } while (left >= DQL_GROUP);
    }
  }
  DQL_SECTION("epilogue");
  DQL_MARK_T(e, 4);
  const StepOut o = period_end(s, mp, e, c, qa, qb, mode);
  DQL_MARK_T(e, 5);
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
  e.code = iv.w & 0xff; e.flags = (iv.w >> 8) & 0xff; e.action = (iv.w >> 16) & 0xff;
  e.reward = T(0.0); e.obs_px = e.obs_vx = e.obs_ax = e.obs_py = e.obs_vy = e.obs_ay = T(0.0);
}
template <typename T> DQL_DEV void store_env(const Env<T>& e, Quad<T>* __restrict__ sr, int4* __restrict__ si, long long n, long long i,
                                             const SimK<T>& c) {
  sr[0 * n + i] = Quad<T>{e.p[0], e.p[1], e.p[2], e.v[0]};
  sr[1 * n + i] = Quad<T>{e.v[1], e.v[2], e.q[0], e.q[1]};
  sr[2 * n + i] = Quad<T>{e.q[2], e.q[3], e.w[0], e.w[1]};
  sr[3 * n + i] = Quad<T>{e.w[2], e.om[0], e.om[1], e.om[2]};
  sr[4 * n + i] = Quad<T>{e.om[3], e.vz_i, e.vz_x1, e.vz_x2};
  sr[5 * n + i] = Quad<T>{e.vz_y1, e.vz_y2, e.vz_y3, e.vz_state};
  sr[6 * n + i] = Quad<T>{e.yw_i, e.yw_x1, e.yw_x2, e.yw_y1};
  sr[7 * n + i] = Quad<T>{e.yw_y2, e.yw_y3, e.yw_state, e.pitch_sp};
  sr[8 * n + i] = Quad<T>{e.mp_phase, e.mp_x, e.mp_u, e.vf_x};
  sr[9 * n + i] = Quad<T>{e.kal_x_x, e.kal_x_P, e.shp_p, e.shp_v};
  sr[10 * n + i] = Quad<T>{e.shp_a, e.cum_x, e.roll_sp, e.mp_y};
  if (c.two_axis || c.traj == DQL_TRAJ_EIGHT) sr[11 * n + i] = Quad<T>{e.mp_v, e.vf_y, e.kal_y_x, e.kal_y_P};
  if (c.two_axis) sr[12 * n + i] = Quad<T>{e.shpy_p, e.shpy_v, e.shpy_a, e.cum_y};
  sr[14 * n + i] = Quad<T>{e.reward, e.obs_px, e.obs_vx, e.obs_ax};
  sr[15 * n + i] = Quad<T>{e.obs_py, e.obs_vy, e.obs_ay, T(0.0)};
  si[i] = make_int4(e.idx_x, e.idx_y, (e.step_count & 0xffff) | (e.cur_check << 16), (e.code & 0xff) | ((e.flags & 0xff) << 8) | ((e.action & 0xff) << 16));
}

}  // namespace dql
