/*
 * dql_oracle.c — CPU restatement (the ORACLE) of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this
 * file.  The product (dql_multirotor_landing_amd + csrc/) never does; it fails loudly without its HIP
 * library.
 *
 * What is restated, and where it comes from (paths relative to /root/reference,
 * pkg = src/dql_multirotor_landing/src/dql_multirotor_landing):
 *   discretise / check / reward / continuous_action / reset   pkg/mdp.py:149-170, 257-569
 *   predict / update / transfer_learning                      pkg/double_q_learning.py:77-146
 *   Kalman, Butterworth                                       pkg/filters.py:4-109
 *   PID.output                                                pkg/pid.py:62-104
 *   SO(3) attitude law + inverse allocation                   pkg/attitude_controller.py:94-156
 *   moving platform                                           pkg/moving_platform.py:87-127
 *   relative observation, acceleration estimate               pkg/observation_utils.py:99-158, 205-268
 *   manager: PID inputs, command mux                          scripts/manager_node.py:192-214, 292-368
 *   rotor force model + first-order rotor filter              src/rotors_simulator/rotors_gazebo_plugins/src/gazebo_motor_model.cpp:358-364, 434-500;
 *                                                             .../include/rotors_gazebo_plugins/common.h:147-183
 *   env reset / step sequencing                               pkg/landing_simulation_env.py:167-282
 * Pinned by: the tests/golden .npz fixtures generated from the reference's own Python modules (tests/golden/make_golden.py).
 * NOT pinned ("parity unpinned"): the rigid-body integrator and contact (Gazebo 11 / ODE are third-party and not
 * under /root/reference, and cannot run anywhere here) — restated as semi-implicit Euler of one rigid body, dt and
 * g from worlds/basic.world:36-73; tf.transformations helpers (third-party) restated from their published formulas.
 *
 * Compiled twice (REAL=float, REAL=double) with -ffp-contract=off: every fused multiply-add is written explicitly
 * as FMA(), so that the arithmetic is reproducible operation by operation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dql.h"

#if ORACLE_F32
typedef float REAL;
#define ORC(name) orc_f32_##name
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define FABS(x) fabsf(x)
#define RINT(x) rintf(x)
#else
typedef double REAL;
#define ORC(name) orc_f64_##name
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(x) sqrt(x)
#define FABS(x) fabs(x)
#define RINT(x) rint(x)
#endif
#define R_(x) ((REAL)(x))
#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * deterministic elementary functions (polynomial kernels after fdlibm's published coefficients;
 * only + - * / fma: identical results wherever IEEE-754 holds)
 * ---------------------------------------------------------------------------------------------- */
static inline REAL det_sin_k(REAL x) {
  const REAL z = x * x;
  REAL r = R_(1.58969099521155010221e-10);
  r = FMA(r, z, R_(-2.50507602534068634195e-08));
  r = FMA(r, z, R_(2.75573137070700676789e-06));
  r = FMA(r, z, R_(-1.98412698298579493134e-04));
  r = FMA(r, z, R_(8.33333333332248946124e-03));
  r = FMA(r, z, R_(-1.66666666666666324348e-01));
  return FMA(x * z, r, x);
}
static inline REAL det_cos_k(REAL x) {
  const REAL z = x * x;
  REAL r = R_(-1.13596475577881948265e-11);
  r = FMA(r, z, R_(2.08757232129817482790e-09));
  r = FMA(r, z, R_(-2.75573143513906633035e-07));
  r = FMA(r, z, R_(2.48015872894767294178e-05));
  r = FMA(r, z, R_(-1.38888888888741095749e-03));
  r = FMA(r, z, R_(4.16666666666666019037e-02));
  return FMA(z * z, r, FMA(z, R_(-0.5), R_(1.0)));
}
static inline void det_sincos(REAL x, REAL* s, REAL* c) {
#if ORACLE_F32
  const REAL pio2_hi = 1.5703125f;                 /* pi/2 to 12 bits  */
  const REAL pio2_lo = 4.8382679489661923e-4f;     /* pi/2 - pio2_hi   */
#else
  const REAL pio2_hi = 1.57079632673412561417e+00; /* pi/2 to 33 bits  */
  const REAL pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi   */
#endif
  const REAL fn = RINT(x * R_(6.36619772367581382433e-01));
  const int n = (int)fn;
  REAL r = FMA(-fn, pio2_hi, x);
  r = FMA(-fn, pio2_lo, r);
  const REAL sk = det_sin_k(r), ck = det_cos_k(r);
  switch (n & 3) {
    case 0: *s = sk; *c = ck; break;
    case 1: *s = ck; *c = -sk; break;
    case 2: *s = -sk; *c = -ck; break;
    default: *s = -ck; *c = sk; break;
  }
}
static inline REAL det_atan(REAL x) {
  static const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                               1.57079632679489655800e+00};
  static const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                               6.12323399573676603587e-17};
  const int neg = x < R_(0.0);
  int id;
  x = FABS(x);
  if (x < R_(0.4375)) {
    id = -1;
  } else if (x < R_(1.1875)) {
    if (x < R_(0.6875)) { id = 0; x = (R_(2.0) * x - R_(1.0)) / (R_(2.0) + x); }
    else { id = 1; x = (x - R_(1.0)) / (x + R_(1.0)); }
  } else if (x < R_(2.4375)) { id = 2; x = (x - R_(1.5)) / (R_(1.0) + R_(1.5) * x); }
  else { id = 3; x = R_(-1.0) / x; }
  const REAL z = x * x, w = z * z;
#if ORACLE_F32 /* round 4: the float kernel's five coefficients (fdlibm s_atanf.c), fused Horner form — csrc/dql_device.hpp det_atan */
  {
    const REAL s1 = z * FMA(w, FMA(w, 6.1687607318e-02f, 1.4253635705e-01f), 3.3333328366e-01f);
    const REAL s2 = w * FMA(w, -1.0648017377e-01f, -1.9999158382e-01f);
    REAL r;
    if (id < 0) r = x - x * (s1 + s2);
    else r = R_(hi[id]) - ((x * (s1 + s2) - R_(lo[id])) - x);
    return neg ? -r : r;
  }
#endif
  const REAL s1 = z * (R_(3.33333333333329318027e-01) + w * (R_(1.42857142725034663711e-01) + w * (R_(9.09088713343650656196e-02) +
                  w * (R_(6.66107313738753120669e-02) + w * (R_(4.97687799461593236017e-02) + w * R_(1.62858201153657823623e-02))))));
  const REAL s2 = w * (R_(-1.99999999998764832476e-01) + w * (R_(-1.11111104054623557880e-01) + w * (R_(-7.69187620504482999495e-02) +
                  w * (R_(-5.83357013379057348645e-02) + w * R_(-3.65315727442169155270e-02)))));
  REAL r;
  if (id < 0) r = x - x * (s1 + s2);
  else r = R_(hi[id]) - ((x * (s1 + s2) - R_(lo[id])) - x);
  return neg ? -r : r;
}
static inline REAL det_atan2(REAL y, REAL x) {
  const REAL pi = R_(3.14159265358979311600e+00), pio2 = R_(1.57079632679489655800e+00);
  if (x == R_(0.0)) {
    if (y == R_(0.0)) return R_(0.0);
    return y > R_(0.0) ? pio2 : -pio2;
  }
  const REAL a = det_atan(FABS(y / x));
  if (x > R_(0.0)) return y < R_(0.0) ? -a : a;
  return y < R_(0.0) ? -(pi - a) : (pi - a);
}
/* natural log for x in (0, 1] (normal numbers) */
static inline REAL det_log(REAL x) {
  int k;
  REAL m;
#if ORACLE_F32
  uint32_t b; memcpy(&b, &x, 4);
  k = (int)(b >> 23) - 127;
  b = (b & 0x007fffffu) | 0x3f800000u; memcpy(&m, &b, 4);
#else
  uint64_t b; memcpy(&b, &x, 8);
  k = (int)(b >> 52) - 1023;
  b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull; memcpy(&m, &b, 8);
#endif
  if (m > R_(1.41421356237309514547e+00)) { m = m * R_(0.5); k += 1; }
  const REAL f = m - R_(1.0);
  const REAL s = f / (R_(2.0) + f);
  const REAL z = s * s, w = z * z;
#if ORACLE_F32 /* round 4: fdlibm e_logf.c's four coefficients, fused — csrc/dql_device.hpp det_log */
  {
    const REAL t1 = w * FMA(w, 0.24279078841f, 0.40000972152f);
    const REAL t2 = z * FMA(w, 0.28498786688f, 0.66666662693f);
    const REAL Rr = t2 + t1, hfsq = R_(0.5) * f * f, dk = (REAL)k;
    return dk * R_(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + Rr) + dk * R_(1.90821492927058770002e-10))) - f);
  }
#endif
  const REAL t1 = w * (R_(3.999999999940941908e-01) + w * (R_(2.222219843214978396e-01) + w * R_(1.531383769920937332e-01)));
  const REAL t2 = z * (R_(6.666666666666735130e-01) + w * (R_(2.857142874366239149e-01) + w * (R_(1.818357216161805012e-01) +
                  w * R_(1.479819860511658591e-01))));
  const REAL Rr = t2 + t1, hfsq = R_(0.5) * f * f, dk = (REAL)k;
  return dk * R_(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + Rr) + dk * R_(1.90821492927058770002e-10))) - f);
}

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 counter RNG (Salmon et al. 2011, published constants)
 * ---------------------------------------------------------------------------------------------- */
static inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define STREAM_ACTION 0u
#define STREAM_INIT 0xFFFFFFFFu
#define STREAM_NOISE0 16u
static inline REAL u24(uint32_t r) { return (REAL)(r >> 8) * R_(5.9604644775390625e-08); }           /* [0,1)  */
static inline REAL u24p(uint32_t r) { return (REAL)((r >> 8) + 1u) * R_(5.9604644775390625e-08); }   /* (0,1]  */
static inline void box_muller(uint32_t ra, uint32_t rb, REAL* n0, REAL* n1) {
  const REAL rad = SQRT(R_(-2.0) * det_log(u24p(ra)));
  REAL s, c;
  det_sincos(R_(6.28318530717958623200e+00) * u24(rb), &s, &c);
  *n0 = rad * c; *n1 = rad * s;
}

/* ------------------------------------------------------------------------------------------------
 * environment record (array of structures here; the HIP product stores structure-of-arrays)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  REAL integ, x1, x2, y1, y2, y3, state; /* integral, Butterworth P-filter history, latched plant state */
  REAL e1, dx1, dx2, dy1, dy2, dy3;      /* previous error + derivative filter history (used when Kd != 0) */
} pid_t_;

typedef struct {
  REAL p[3], v[3], q[4], w[3], om[4];
  pid_t_ vz, yaw;
  REAL pitch_sp, roll_sp;
  REAL mp_phase, mp_x, mp_y, mp_u, mp_v, mp_r, mp_w;
  REAL vf[2], kal_x[2], kal_P[2];
  REAL shp[2][3], cum[2];
  REAL reward, obs[6], pad[3];
  int32_t idx[2], step_count, cur_check, code, flags, action;
} env_t;

enum { FL_DONE = 1, FL_CONTACT = 2, FL_ACC_INIT = 4, FL_WAS_RESET = 8, FL_OBS_CONTACT = 16 };

/* Field order = the product's HBM layout: groups of four consecutive fields form one 16/32-byte "quad" per env
 * (quads 0-10 x-axis state, 11-12 y-axis state, 13 per-env platform constants, 14-15 outputs). */
#define NF_REAL 64
#define NF_INT 7
static const char* const k_real_names[NF_REAL] = {
    "px", "py", "pz", "vx", "vy", "vz", "qw", "qx", "qy", "qz", "wx", "wy", "wz", "om0", "om1", "om2",
    "om3", "vz_i", "vz_x1", "vz_x2", "vz_y1", "vz_y2", "vz_y3", "vz_state",
    "yw_i", "yw_x1", "yw_x2", "yw_y1", "yw_y2", "yw_y3", "yw_state", "pitch_sp",
    "mp_phase", "mp_x", "mp_u", "vf_x", "kal_x_x", "kal_x_P", "shp_x_p", "shp_x_v",
    "shp_x_a", "cum_x", "roll_sp", "mp_y", "mp_v", "vf_y", "kal_y_x", "kal_y_P",
    "shp_y_p", "shp_y_v", "shp_y_a", "cum_y", "mp_r", "mp_w", "pad0", "pad1",
    "reward", "obs_p_x", "obs_v_x", "obs_a_x", "obs_p_y", "obs_v_y", "obs_a_y", "pad2"};
static const char* const k_int_names[NF_INT] = {"idx_x", "idx_y", "step_count", "cur_check", "code", "flags", "action"};

#define FIELD_MAP(X, e)                                                                                                     \
  X(e->p[0]) X(e->p[1]) X(e->p[2]) X(e->v[0]) X(e->v[1]) X(e->v[2]) X(e->q[0]) X(e->q[1]) X(e->q[2]) X(e->q[3])          \
  X(e->w[0]) X(e->w[1]) X(e->w[2]) X(e->om[0]) X(e->om[1]) X(e->om[2]) X(e->om[3])                                       \
  X(e->vz.integ) X(e->vz.x1) X(e->vz.x2) X(e->vz.y1) X(e->vz.y2) X(e->vz.y3) X(e->vz.state)                              \
  X(e->yaw.integ) X(e->yaw.x1) X(e->yaw.x2) X(e->yaw.y1) X(e->yaw.y2) X(e->yaw.y3) X(e->yaw.state) X(e->pitch_sp)        \
  X(e->mp_phase) X(e->mp_x) X(e->mp_u) X(e->vf[0]) X(e->kal_x[0]) X(e->kal_P[0]) X(e->shp[0][0]) X(e->shp[0][1])          \
  X(e->shp[0][2]) X(e->cum[0]) X(e->roll_sp) X(e->mp_y) X(e->mp_v) X(e->vf[1]) X(e->kal_x[1]) X(e->kal_P[1])              \
  X(e->shp[1][0]) X(e->shp[1][1]) X(e->shp[1][2]) X(e->cum[1]) X(e->mp_r) X(e->mp_w) X(e->pad[0]) X(e->pad[1])            \
  X(e->reward) X(e->obs[0]) X(e->obs[2]) X(e->obs[4]) X(e->obs[1]) X(e->obs[3]) X(e->obs[5]) X(e->pad[2])

static void env_to_fields(const env_t* e, double* f, int32_t* g) {
  int i = 0;
#define X(member) f[i++] = (double)(member);
  FIELD_MAP(X, e)
#undef X
  g[0] = e->idx[0]; g[1] = e->idx[1]; g[2] = e->step_count; g[3] = e->cur_check; g[4] = e->code; g[5] = e->flags; g[6] = e->action;
}
static void fields_to_env(env_t* e, const double* f, const int32_t* g) {
  int i = 0;
#define X(member) (member) = (REAL)f[i++];
  FIELD_MAP(X, e)
#undef X
  e->idx[0] = g[0]; e->idx[1] = g[1]; e->step_count = g[2]; e->cur_check = g[3]; e->code = g[4]; e->flags = g[5]; e->action = g[6];
}

/* ------------------------------------------------------------------------------------------------
 * MDP — pkg/mdp.py
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  REAL p_max, v_max, a_max, theta_max, delta_theta, beta, sigma_a, min_alt;
  REAL w_p, w_v, w_theta, w_dur, w_fail, w_succ, delta_t, f_ag, timeout_steps;
  REAL lim_p[5], lim_v[5], lim_a[5], angles[7];
  REAL inv_p_max, inv_v_max, inv_a_max, inv_theta_max, dtheta_ratio; /* float32 MDP: host reciprocals (round 4) */
  REAL tan2_mid[3]; /* float32 fused step (round 5): tan^2 of the angle grid's bin boundaries (angle_bin_from_tangent) */
  int working, goal_logic;
  uint32_t quirks;
} mdpc_t;

static void mdpc_init(mdpc_t* m, const dql_config* c) {
  m->p_max = (REAL)c->p_max; m->v_max = (REAL)c->v_max; m->a_max = (REAL)c->a_max;
  m->theta_max = (REAL)c->theta_max; m->delta_theta = (REAL)c->delta_theta; m->beta = (REAL)c->beta; m->sigma_a = (REAL)c->sigma_a;
  m->min_alt = (REAL)c->minimum_altitude;
  m->w_p = (REAL)c->w_p; m->w_v = (REAL)c->w_v; m->w_theta = (REAL)c->w_theta; m->w_dur = (REAL)c->w_dur; m->w_fail = (REAL)c->w_fail; m->w_succ = (REAL)c->w_succ;
  m->delta_t = (REAL)(1.0 / c->f_ag); /* pkg/mdp.py:147 */
  m->f_ag = (REAL)c->f_ag;
  m->timeout_steps = (REAL)(c->t_max * c->f_ag); /* pkg/mdp.py:395 */
  for (int i = 0; i < 5; ++i) { m->lim_p[i] = (REAL)c->lim_p[i]; m->lim_v[i] = (REAL)c->lim_v[i]; m->lim_a[i] = (REAL)c->lim_a[i]; }
  /* np.linspace(-theta_max, theta_max, 7): i*step + start, last element = stop  (pkg/mdp.py:145) */
  const double step = (c->theta_max - (-c->theta_max)) / 6.0;
  for (int i = 0; i < 6; ++i) m->angles[i] = (REAL)((double)i * step + (-c->theta_max));
  m->angles[6] = (REAL)c->theta_max;
  m->inv_p_max = (REAL)(1.0 / c->p_max); m->inv_v_max = (REAL)(1.0 / c->v_max); m->inv_a_max = (REAL)(1.0 / c->a_max);
  m->inv_theta_max = (REAL)(1.0 / c->theta_max); m->dtheta_ratio = (REAL)(c->delta_theta / c->theta_max);
  for (int j = 0; j < 3; ++j) { const double t = tan(((double)j + 0.5) * step); m->tan2_mid[j] = (REAL)(t * t); }
  m->working = c->working_curriculum_step; m->goal_logic = c->goal_logic;
  m->quirks = c->quirks;
}
/* FLOAT32 MDP, ROUND 4: the normalising divisions by the constants p_max, v_max, a_max, theta_max are multiplications by the host's
 * reciprocals (double division rounded to float once), as in the float32 kernel (csrc/dql_device.hpp, Fast32).  float64 keeps the
 * reference's divisions: it is what G1 / G2 pin. */
#if ORACLE_F32
#define NORM(x, d, inv) ((x) * (inv))
#else
#define NORM(x, d, inv) ((x) / (d))
#endif
static inline REAL clip(REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* The product's 500 Hz clips are one v_med3_f32 each in float (csrc/dql_device.hpp: clip3); restated here from the gfx9 ISA
 * pseudo-code so that even signed-zero ties agree: V_MAX_F32 orders -0 < +0, V_MED3_F32 returns the max of the two operands
 * that are not the maximum (min3 when an operand is NaN).  Double keeps the comparison chain, as the product does. */
#if ORACLE_F32
static inline float amd_max_f32(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return (signbit(a) && signbit(b)) ? a : 0.0f;
  return a >= b ? a : b;
}
static inline float amd_min_f32(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return (signbit(a) || signbit(b)) ? -0.0f : a;
  return a <= b ? a : b;
}
static inline REAL clip3(REAL a, REAL b, REAL c) {
  if (a != a || b != b || c != c) return amd_min_f32(amd_min_f32(a, b), c);
  const float mx = amd_max_f32(amd_max_f32(a, b), c);
  if (mx == a) return amd_max_f32(b, c);
  if (mx == b) return amd_max_f32(a, c);
  return amd_max_f32(a, b);
}
#else
static inline REAL clip3(REAL x, REAL lo, REAL hi) { return clip(x, lo, hi); }
#endif

/* pkg/mdp.py:149-158 */
static inline int latest_valid_level(const REAL* lim, int n, REAL value) {
  for (int idx = 1; idx < n; ++idx) {
    if (value < -lim[idx] || value > lim[idx]) return idx - 1;
  }
  return n - 1;
}
/* pkg/mdp.py:160-170; -1 where the reference raises ValueError */
static inline int disc3(REAL v, REAL goal, REAL limit) {
  if (-limit <= v && v < -goal) return 0;
  if (-goal <= v && v <= goal) return 1;
  if (v <= limit) return 2;
  return -1;
}
/* pkg/mdp.py:257-333 -> packed index ((((k*3+p)*3+v)*3+a)*7+theta), or -1 */
/* bin_given >= 0: the angle's grid bin comes from angle_bin_from_tangent (float32 fused step), the angle argument is not read */
static int discretise_bin(const mdpc_t* m, REAL rel_p, REAL rel_v, REAL rel_a, REAL angle, int bin_given);
static int discretise(const mdpc_t* m, REAL rel_p, REAL rel_v, REAL rel_a, REAL angle) { return discretise_bin(m, rel_p, rel_v, rel_a, angle, -1); }
#if ORACLE_F32
/* csrc/dql_device.hpp angle_bin_from_tangent: the nearest grid angle of atan2(s, c), c^2 = c2, from three comparisons of s^2 with tan^2(boundary) c^2 */
static int angle_bin_from_tangent(const mdpc_t* m, REAL s, REAL c2, int c_pos) {
  const REAL s2 = s * s;
  const REAL t0 = m->tan2_mid[0] * c2, t1 = m->tan2_mid[1] * c2, t2 = m->tan2_mid[2] * c2;
  const int neg = s < R_(0.0);
  const int b0 = neg ? s2 >= t0 : s2 > t0, b1 = neg ? s2 >= t1 : s2 > t1, b2 = neg ? s2 >= t2 : s2 > t2;
  int c = b2 ? 3 : (b1 ? 2 : (b0 ? 1 : 0));
  if (!c_pos) c = (s != R_(0.0)) ? 3 : 0;
  return neg ? 3 - c : 3 + c;
}
#endif
static int discretise_bin(const mdpc_t* m, REAL rel_p, REAL rel_v, REAL rel_a, REAL angle, int bin_given) {
  const REAL cp = clip(NORM(rel_p, m->p_max, m->inv_p_max), R_(-1.0), R_(1.0));
  const REAL cv = clip(NORM(rel_v, m->v_max, m->inv_v_max), R_(-1.0), R_(1.0));
  const REAL ca = clip(NORM(rel_a, m->a_max, m->inv_a_max), R_(-1.0), R_(1.0));
  const int n = m->working + 1;
  int k = latest_valid_level(m->lim_p, n, cp);
  const int kv = latest_valid_level(m->lim_v, n, cv), ka = latest_valid_level(m->lim_a, n, ca);
  if (kv < k) k = kv;
  if (ka < k) k = ka;
  REAL pc = m->beta, vc = m->beta, ac = m->sigma_a;
  if (k < m->working) { pc = m->lim_p[k + 1] / m->lim_p[k]; vc = m->lim_v[k + 1] / m->lim_v[k]; }
  if (k == m->working) ac = ac * m->beta;
  const int dp = disc3(cp, m->lim_p[k] * pc, m->lim_p[k]);
  const int dv = disc3(cv, m->lim_v[k] * vc, m->lim_v[k]);
  const int da = disc3(ca, m->lim_a[k] * ac, m->lim_a[k]);
  if (dp < 0 || dv < 0 || da < 0) return -1;
  int best = bin_given;
  if (bin_given < 0) {
    const REAL ct = clip(angle, -m->theta_max, m->theta_max);
    best = 0; REAL bd = FABS(m->angles[0] - ct);
    for (int i = 1; i < 7; ++i) { const REAL d = FABS(m->angles[i] - ct); if (d < bd) { bd = d; best = i; } } /* np.argmin: first minimum */
  }
  return (((k * 3 + dp) * 3 + dv) * 3 + da) * 7 + best;
}
static inline int idx_level(int idx) { return idx / DQL_STATES_PER_LEVEL; }
static inline int idx_pos(int idx) { return (idx / 63) % 3; }
static inline int idx_vel(int idx) { return (idx / 21) % 3; }

/* pkg/mdp.py:543-560 */
static inline REAL continuous_action(const mdpc_t* m, REAL sp, int action) {
  if (action == 0) { const REAL t = sp + m->delta_theta; return t < m->theta_max ? t : m->theta_max; }
  if (action == 1) { const REAL t = sp - m->delta_theta; return t > -m->theta_max ? t : -m->theta_max; }
  return sp;
}
/* pkg/mdp.py:335-439.  Returns the (possibly sticky) check code. */
static int mdp_check(const mdpc_t* m, int* step_count, int* cur_check, int code, int prev_idx, int cur_idx, int contact,
                     REAL rel_p_x, REAL rel_p_y, REAL abs_p_z, int two, int prev_idy, int cur_idy) {
  /* two-axis configs (beyond the reference, B16): the goal state is the joint goal of both 1-D MDPs */
  const int goal_x = prev_idx >= 0 && idx_pos(cur_idx) == 1 && idx_vel(cur_idx) == 1;
  const int goal_y = !two || (prev_idy >= 0 && idx_pos(cur_idy) == 1 && idx_vel(cur_idy) == 1);
  const int lvl_x = idx_level(prev_idx) == m->working && idx_level(cur_idx) == m->working;
  const int lvl_y = !two || (idx_level(prev_idy) == m->working && idx_level(cur_idy) == m->working);
  *step_count += 1;
  if (!(m->quirks & DQL_Q_STICKY_CHECK)) code = DQL_NON_TERMINAL;
  if (contact) code = DQL_TERMINAL_CONTACT;
  else if (rel_p_x < -m->p_max || rel_p_x > m->p_max) code = DQL_TERMINAL_FLYZONE_X;
  else if (rel_p_y < -m->p_max || rel_p_y > m->p_max) code = DQL_TERMINAL_FLYZONE_Y;
  else if (abs_p_z < m->min_alt) code = DQL_TERMINAL_MINIMUM_ALTITUDE;
  else if (abs_p_z > m->p_max) code = DQL_TERMINAL_FLYZONE_Z;
  else if ((REAL)*step_count >= m->timeout_steps) code = DQL_TERMINAL_TIMEOUT;
  else if (m->goal_logic && goal_x && goal_y) {
    if (lvl_x && lvl_y) {
      *cur_check += 1;
      code = ((REAL)*cur_check >= m->f_ag) ? DQL_TERMINAL_SUCCESS : DQL_NON_TERMINAL_SUCCESS;
    } else {
      *cur_check = 0;
    }
  } else if (!(m->quirks & DQL_Q_GOAL_COUNT_KEPT)) {
    *cur_check = 0; /* strict variant: leaving the goal bins loses the progress (the reference keeps it, pkg/mdp.py:402-425) */
  }
  return code;
}
/* pkg/mdp.py:441-541; shp = persistent "current_shaping_value" (position, velocity, angle) */
static REAL mdp_reward(const mdpc_t* m, REAL* shp, REAL* cum, int code, int cur_idx, REAL rel_p, REAL rel_v, REAL angle_sp) {
  const REAL ncp = clip(NORM(rel_p, m->p_max, m->inv_p_max), R_(-1.0), R_(1.0));
  const REAL ncv = clip(NORM(rel_v, m->v_max, m->inv_v_max), R_(-1.0), R_(1.0));
  const REAL npitch = NORM(angle_sp, m->theta_max, m->inv_theta_max);
  const int k = idx_level(cur_idx);
  const REAL prev_p = shp[0], prev_v = shp[1], prev_a = shp[2];
  shp[0] = m->w_p * FABS(ncp); shp[1] = m->w_v * FABS(ncv); shp[2] = m->w_theta * FABS(npitch);
  const REAL r_p_max = FABS(m->w_p) * m->lim_v[k] * m->delta_t;
  const REAL r_v_max = FABS(m->w_v) * m->lim_a[k] * m->delta_t;
#if ORACLE_F32
  const REAL r_theta_max = FABS(m->w_theta) * m->dtheta_ratio * m->lim_v[k];
#else
  const REAL r_theta_max = FABS(m->w_theta) * (m->delta_theta / m->theta_max) * m->lim_v[k];
#endif
  const REAL r_dur_max = m->w_dur * m->lim_v[k] * m->delta_t;
  const REAL r_max = r_p_max + r_v_max + r_theta_max + r_dur_max;
  const REAL r_p = clip(shp[0] - prev_p, -r_p_max, r_p_max);
  const REAL r_v = clip(shp[1] - prev_v, -r_v_max, r_v_max);
  const REAL r_theta = NORM(m->w_theta * (FABS(shp[2]) - FABS(prev_a)), m->theta_max, m->inv_theta_max) * m->lim_v[k];
  const REAL r_dur = m->w_dur * m->lim_v[k] * m->delta_t;
  REAL r_term;
  if (code == DQL_NON_TERMINAL_SUCCESS || code == DQL_TERMINAL_SUCCESS) r_term = m->w_succ * r_max;
  else if (code == DQL_NON_TERMINAL && !(m->quirks & DQL_Q_FAIL_TERM_EVERY_STEP)) r_term = R_(0.0);
  else r_term = m->w_fail * r_max;
  const REAL r_t = r_p + r_v + r_theta + r_dur + r_term;
  *cum += r_t;
  return r_t;
}

/* ------------------------------------------------------------------------------------------------
 * agent — pkg/double_q_learning.py (tables are float64 whatever REAL is)
 * ---------------------------------------------------------------------------------------------- */
/* double -> int64 fixed point: round to nearest even, saturating at +-2^50 (csrc/dql_device.hpp fx_round: the kernel rounds with the 2^52 trick, which is
 * llrint() on that range) */
static inline long long fx_round(double x) { return llrint(fmin(fmax(x, -0x1p50), 0x1p50)); }
static inline int argmax3(double a, double b, double c) { int k = 0; double v = a; if (b > v) { v = b; k = 1; } if (c > v) { k = 2; } return k; }
/* :119-124 */
static inline int agent_predict(const double* qa, const double* qb, int idx) {
  const double* a = qa + idx * 3; const double* b = qb + idx * 3;
  return argmax3((a[0] + b[0]) / 2, (a[1] + b[1]) / 2, (a[2] + b[2]) / 2);
}

/* ------------------------------------------------------------------------------------------------
 * filters / PID — pkg/filters.py, pkg/pid.py
 * ---------------------------------------------------------------------------------------------- */
typedef struct { REAL k1, k2, inv_denom, b2, a2, a3; } bwc_t;
static void bwc_init(bwc_t* b, double c) {
  const double denom = 1 + c * c + 1.414 * c;             /* pkg/filters.py:94 */
  b->inv_denom = (REAL)(1.0 / denom);                      /* :103 */
  b->k1 = (REAL)(c * c - 1.414 * c + 1);                   /* :105 */
  b->k2 = (REAL)(-2 * c * c + 2);                          /* :106 */
  /* float32 (round 4): the same transfer function b (1 + 2 z^-1 + z^-2) / (1 + a2 z^-2 + a3 z^-3) in TRANSPOSED form */
  b->b2 = (REAL)(2.0 / denom); b->a2 = (REAL)((-2 * c * c + 2) / denom); b->a3 = (REAL)((c * c - 1.414 * c + 1) / denom);
}
/* pkg/filters.py:98-109; history x1,x2 = previous two inputs, y1..y3 = previous three outputs (the deque keeps 3) */
static inline REAL butterworth(const bwc_t* b, REAL x0, REAL* x1, REAL* x2, REAL* y1, REAL* y2, REAL* y3) {
  /* FLOAT32 TICK, ROUND 3 (csrc/dql_device.hpp, Fast32): the float64 build spells the reference's expressions out operation by
   * operation — it is what the golden vectors pin.  The float32 build takes the same formulas in their shortest
   * correctly-rounded-per-operation form (products folded into the additions that consume them, dt/m, dt g, dt/I multiplied out on
   * the host, two Newton steps from a second-order start in yaw_cs), exactly as the float32 kernel does: ORACLE_F32 below. */
#if ORACLE_F32
  /* FLOAT32 TICK, ROUND 4: y[n] = b (x[n] + 2 x[n-1] + x[n-2]) - a2 y[n-2] - a3 y[n-3] (the reference's recurrence, deque of three included)
   * as a transposed direct form: y = b x + t1; t1 = 2b x + t2; t2 = b x - a2 y + t3; t3 = -a3 y.  Three states instead of five histories and no
   * history shifts (ten register moves per physics tick in the kernel's rolled loop).  States: t1, t2, t3 live in the fields x1, x2, y1; y2, y3
   * are not used.  The reference's histories (x1, x2 = last two inputs, y1..y3 = last three outputs) map to them as t1 = 2b x1 + b x2 - a2 y2 - a3 y3,
   * t2 = b x1 - a2 y1 - a3 y2, t3 = -a3 y1 (tests/test_gpu_parity.py: to_f32_filter_state). */
  (void)y2; (void)y3;
  const REAL value = FMA(b->inv_denom, x0, *x1);
  *x1 = FMA(b->b2, x0, *x2);
  *x2 = FMA(b->inv_denom, x0, *y1);
  if (b->k2 != R_(0.0)) *x2 = FMA(-b->a2, value, *x2);
  *y1 = -(b->a3 * value);
  return value;
#else
  REAL acc = *x2 + R_(2.0) * *x1 + x0 - b->k1 * *y3;
  if (b->k2 != R_(0.0)) acc = acc - (b->k2 * *y2); /* -2c^2 + 2 is exactly 0 for the reference's c = 1 (pkg/filters.py:93,106) */
  const REAL value = b->inv_denom * acc;
  *x2 = *x1; *x1 = x0;
  *y3 = *y2; *y2 = *y1; *y1 = value;
  return value;
#endif
}
typedef struct { REAL kp, ki, kd, lo, hi, windup, setpoint; } pidc_t;
/* pkg/pid.py:62-104 with delta_t > 0 */
static inline REAL pid_output(const pidc_t* c, const bwc_t* b, pid_t_* s, REAL delta_t) {
  const REAL e0 = c->setpoint - s->state;
#if ORACLE_F32
  s->integ = clip3(FMA(e0, delta_t, s->integ), -c->windup, c->windup);
#else
  s->integ = clip3(s->integ + e0 * delta_t, -c->windup, c->windup);
#endif
  const REAL fe = butterworth(b, e0, &s->x1, &s->x2, &s->y1, &s->y2, &s->y3);
#if ORACLE_F32
  REAL eff = FMA(c->kp, fe, c->ki * s->integ);
#else
  REAL eff = c->kp * fe + c->ki * s->integ;
#endif
  if (c->kd != R_(0.0)) {
    const REAL draw = (e0 - s->e1) / delta_t;
    const REAL fd = butterworth(b, draw, &s->dx1, &s->dx2, &s->dy1, &s->dy2, &s->dy3);
    eff = eff + c->kd * fd;
  }
  s->e1 = e0;
  return clip3(eff, c->lo, c->hi);
}
/* pkg/filters.py:19-36 */
static inline REAL kalman1d(REAL* x, REAL* P, REAL Q, REAL Rm, REAL z) {
  *P += Q;
  const REAL K = (Rm == R_(0.0)) ? R_(1.0) : *P / (*P + Rm); /* P / (P + 0) is exactly 1 */
  *x += K * (z - *x);
  *P *= (R_(1.0) - K);
  return *x;
}

/* ------------------------------------------------------------------------------------------------
 * simulator constants
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  REAL dt, g, inv_m, I[3], inv_I[3], l, h, kf, km, lkf, kmkf, aup, adn, omax, cd, crd;
  REAL dtm, dtg, dtI[3]; /* dt / m, dt g, dt / I (float32 tick) */
  REAL oup, odn, inv_mgr_dt; /* 1 - rotor alpha, 1 / manager period (float32 tick) */
  REAL nlcd, hdt, low_z; /* -(l c_d), dt / 2, mp_top + bottom (float32 tick, round 4b) */
  REAL kR[3], kW[3], ia, ib, ic; /* inverse allocation coefficients */
  pidc_t pvz, pyaw; bwc_t bw;
  REAL mp_dt, mp_top, mp_hx, mp_hy, bottom, z_init, init_sigma, p_max;
  REAL noise_p, noise_v, kal_q, kal_r, mgr_dt;
  int div, traj, init_uniform, working, per_env_platform, two_axis;
  REAL mp_r, mp_w, mp_r_lo, mp_r_hi, mp_t_lo, mp_t_hi;
  uint32_t quirks;
} simc_t;

static void simc_init(simc_t* s, const dql_config* c) {
  s->dt = (REAL)c->dt; s->g = (REAL)c->gravity; s->inv_m = (REAL)(1.0 / c->mass);
  s->dtm = (REAL)(c->dt / c->mass); s->dtg = (REAL)(c->dt * c->gravity);
  for (int i = 0; i < 3; ++i) s->dtI[i] = (REAL)(c->dt / c->inertia[i]);
  s->nlcd = (REAL)(-(c->arm_length * c->c_drag)); s->hdt = (REAL)(0.5 * c->dt); s->low_z = (REAL)(c->mp_top_z + c->drone_bottom);
  s->oup = (REAL)(1.0 - c->rotor_alpha_up); s->odn = (REAL)(1.0 - c->rotor_alpha_down); s->inv_mgr_dt = (REAL)(1.0 / (c->dt * c->manager_div));
  for (int i = 0; i < 3; ++i) { s->I[i] = (REAL)c->inertia[i]; s->inv_I[i] = (REAL)(1.0 / c->inertia[i]); s->kR[i] = (REAL)c->k_R[i]; s->kW[i] = (REAL)c->k_W[i]; }
  s->l = (REAL)c->arm_length; s->h = (REAL)c->rotor_z; s->kf = (REAL)c->k_f; s->km = (REAL)c->k_m;
  s->lkf = (REAL)(c->arm_length * c->k_f); s->kmkf = (REAL)(c->k_m * c->k_f);
  s->aup = (REAL)c->rotor_alpha_up; s->adn = (REAL)c->rotor_alpha_down; s->omax = (REAL)c->rotor_max;
  s->cd = (REAL)c->c_drag; s->crd = (REAL)(c->c_roll / c->c_drag);
  /* closed-form inverse of the allocation matrix of pkg/attitude_controller.py:94-104 ("plus" layout) */
  s->ia = (REAL)(1.0 / (4.0 * c->k_f)); s->ib = (REAL)(1.0 / (2.0 * c->arm_length * c->k_f)); s->ic = (REAL)(1.0 / (4.0 * c->k_f * c->k_m));
  const double* pv = c->pid_vz; const double* py = c->pid_yaw;
  s->pvz = (pidc_t){(REAL)pv[0], (REAL)pv[1], (REAL)pv[2], (REAL)pv[3], (REAL)pv[4], (REAL)pv[5], (REAL)c->vz_setpoint};
  s->pyaw = (pidc_t){(REAL)py[0], (REAL)py[1], (REAL)py[2], (REAL)py[3], (REAL)py[4], (REAL)py[5], (REAL)c->yaw_setpoint};
  bwc_init(&s->bw, c->bw_c);
  s->mp_dt = (REAL)c->mp_dt; s->mp_top = (REAL)c->mp_top_z; s->mp_hx = (REAL)c->mp_half_x; s->mp_hy = (REAL)c->mp_half_y;
  s->bottom = (REAL)c->drone_bottom; s->z_init = (REAL)c->z_init; s->init_sigma = (REAL)c->init_sigma; s->p_max = (REAL)c->p_max;
  s->noise_p = (REAL)c->noise_pos_sd; s->noise_v = (REAL)c->noise_vel_sd; s->kal_q = (REAL)c->kalman_q;
  s->kal_r = (REAL)(c->noise_vel_sd * c->noise_vel_sd); /* pkg/filters.py:49 */
  s->mgr_dt = (REAL)(c->dt * c->manager_div);
  s->div = c->manager_div; s->traj = c->trajectory; s->init_uniform = c->init_uniform; s->working = c->working_curriculum_step;
  s->per_env_platform = c->per_env_platform; s->two_axis = c->two_axis;
  s->mp_r = (REAL)c->mp_r_x; s->mp_w = (REAL)(c->mp_t_x / c->mp_r_x);
  if (c->trajectory == DQL_TRAJ_EIGHT) { s->mp_r = R_(3.0); s->mp_w = (REAL)(0.8 / 3.0); } /* pkg/moving_platform.py:93-97 */
  s->mp_r_lo = (REAL)c->mp_r_lo; s->mp_r_hi = (REAL)c->mp_r_hi; s->mp_t_lo = (REAL)c->mp_t_lo; s->mp_t_hi = (REAL)c->mp_t_hi;
  s->quirks = c->quirks;
}

/* rotation matrix of a unit quaternion (w, x, y, z) */
static inline void quat_to_R(const REAL* q, REAL R[9]) {
  const REAL w = q[0], x = q[1], y = q[2], z = q[3];
#if ORACLE_F32 /* round 4: the factor 2 applied once to x, y, z (exact), every entry one or two fused multiply-adds: 17 operations instead of 30 */
  const REAL x2 = x + x, y2 = y + y, z2 = z + z;
  const REAL t = FMA(-z, z2, R_(1.0));
  R[0] = FMA(-y, y2, t); R[4] = FMA(-x, x2, t); R[8] = FMA(-x, x2, FMA(-y, y2, R_(1.0)));
  const REAL wx2 = w * x2, wy2 = w * y2, wz2 = w * z2;
  R[1] = FMA(x, y2, -wz2); R[3] = FMA(x, y2, wz2);
  R[2] = FMA(x, z2, wy2); R[6] = FMA(x, z2, -wy2);
  R[5] = FMA(y, z2, -wx2); R[7] = FMA(y, z2, wx2);
  return;
#endif
  const REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  R[0] = R_(1.0) - R_(2.0) * (yy + zz); R[1] = R_(2.0) * (xy - wz); R[2] = R_(2.0) * (xz + wy);
  R[3] = R_(2.0) * (xy + wz); R[4] = R_(1.0) - R_(2.0) * (xx + zz); R[5] = R_(2.0) * (yz - wx);
  R[6] = R_(2.0) * (xz - wy); R[7] = R_(2.0) * (yz + wx); R[8] = R_(1.0) - R_(2.0) * (xx + yy);
}
/* cos/sin of yaw = atan2(R10, R00) without the angle (pkg/attitude_controller.py:136-137) */
/* n2 = cos^2(tilt) is close to 1 in flight: 1/sqrt(n2) by Newton's iteration from r0 = 1.5 - 0.5 n2 (multiplies and fmas
 * only; converged to rounding for tilt < ~55 deg, degrades gracefully, never NaN, for a tumbling vehicle) */
static inline void yaw_cs4(const REAL R[9], REAL* c, REAL* s, REAL* ct, REAL* rn) {
  const REAL n2 = FMA(R[0], R[0], R[3] * R[3]);
  const REAL h = R_(-0.5) * n2;
#if ORACLE_F32 /* second-order start 1 + d/2 + 3 d^2/8, d = 1 - n2, then THREE Newton steps (round 4; two left 0.5 % at a tilt of 60 deg) */
  const REAL d = R_(1.0) - n2;
  REAL r = FMA(FMA(R_(0.375), d, R_(0.5)), d, R_(1.0));
  for (int k = 0; k < 3; ++k) r = r * FMA(h * r, r, R_(1.5));
#else
  REAL r = FMA(R_(-0.5), n2, R_(1.5));
  for (int k = 0; k < 5; ++k) r = r * FMA(h * r, r, R_(1.5));
#endif
  *c = R[0] * r; *s = R[3] * r; *ct = n2 * r; *rn = r; /* ct = cos(tilt), rn = 1 / cos(tilt): the float32 attitude law's yaw-free attitude */
}
static inline void yaw_cs(const REAL R[9], REAL* c, REAL* s) { REAL ct, rn; yaw_cs4(R, c, s, &ct, &rn); }

/* pkg/attitude_controller.py:107-156: (R, body rates, B = Rx(roll_sp) Ry(pitch_sp), yaw rate cmd, thrust) -> rotor speed commands */
/* xonly (float32, round 4): the caller flies an x-axis config — roll set-point exactly 0, B = Ry(pitch_sp) — and the float32 kernel then takes the
 * collapsed form of E (csrc/dql_device.hpp attitude(): E10 = 0, E12 = -sin(roll), no middle terms) */
static inline void attitude(const simc_t* s, const REAL R[9], const REAL w[3], const REAL B[9], REAL cy, REAL sy, REAL ct, REAL rn, REAL r_cmd, REAL thrust,
                            REAL cmd[4], REAL M[3], int xonly) {
#if ORACLE_F32
  /* float32 tick: E = R_des^T R = B^T A on the yaw-free attitude A = Ry(pitch) Rx(roll): last row = R's, A00 = ct, A10 = 0,
   * sin / cos(roll) = R7 rn, R8 rn, sin(pitch) = -R6 (csrc/dql_device.hpp attitude()) */
  (void)cy; (void)sy;
  const REAL sr = R[7] * rn, cr = R[8] * rn;
  const REAL A01 = -(R[6] * sr), A02 = -(R[6] * cr);
  REAL E01, E02, E10, E12, E20, E21, E22;
  if (xonly) {
    const REAL cp = B[0], sp = B[2];
    E01 = FMA(cp, A01, -(sp * R[7])); E02 = FMA(cp, A02, -(sp * R[8]));
    E10 = R_(0.0); E12 = -sr;
    E20 = FMA(sp, ct, cp * R[6]); E21 = FMA(sp, A01, cp * R[7]); E22 = FMA(sp, A02, cp * R[8]);
  } else {
    E01 = FMA(B[0], A01, FMA(B[3], cr, B[6] * R[7]));
    E02 = FMA(B[0], A02, FMA(B[3], -sr, B[6] * R[8]));
    E10 = FMA(B[1], ct, B[7] * R[6]);
    E12 = FMA(B[1], A02, FMA(B[4], -sr, B[7] * R[8]));
    E20 = FMA(B[2], ct, B[8] * R[6]);
    E21 = FMA(B[2], A01, FMA(B[5], cr, B[8] * R[7]));
    E22 = FMA(B[2], A02, FMA(B[5], -sr, B[8] * R[8]));
  }
#else
  (void)ct; (void)rn; (void)xonly;
  REAL D[9]; /* R_des = Rz(yaw) B */
  for (int j = 0; j < 3; ++j) { D[j] = FMA(cy, B[j], -(sy * B[3 + j])); D[3 + j] = FMA(sy, B[j], cy * B[3 + j]); D[6 + j] = B[6 + j]; }
#define E_(i, j) FMA(D[i], R[j], FMA(D[3 + i], R[3 + j], D[6 + i] * R[6 + j])) /* (R_des^T R)_ij */
  const REAL E01 = E_(0, 1), E10 = E_(1, 0), E02 = E_(0, 2), E20 = E_(2, 0), E12 = E_(1, 2), E21 = E_(2, 1), E22 = E_(2, 2);
#undef E_
#endif
#if ORACLE_F32 /* xonly: the kernel forms -E01 directly (E10 - E01 with E10 = +0 would turn a zero E01 into +0 instead of -0) */
  const REAL eR0 = R_(0.5) * (E21 - E12), eR1 = R_(0.5) * (E02 - E20), eR2 = R_(0.5) * (xonly ? -E01 : E10 - E01);
#else
  const REAL eR0 = R_(0.5) * (E21 - E12), eR1 = R_(0.5) * (E02 - E20), eR2 = R_(0.5) * (E10 - E01);
#endif
#if ORACLE_F32
  const REAL eW0 = FMA(-r_cmd, E02, w[0]), eW1 = FMA(-r_cmd, E12, w[1]), eW2 = FMA(-r_cmd, E22, w[2]);
  M[0] = FMA(-eW0, s->kW[0], -(eR0 * s->kR[0]));
  M[1] = FMA(-eW1, s->kW[1], -(eR1 * s->kR[1]));
  M[2] = FMA(-eW2, s->kW[2], -(eR2 * s->kR[2]));
  const REAL a = thrust * s->ia;
  const REAL w2[4] = {FMA(M[2], s->ic, FMA(-M[1], s->ib, a)), FMA(-M[2], s->ic, FMA(M[0], s->ib, a)), FMA(M[2], s->ic, FMA(M[1], s->ib, a)),
                      FMA(-M[2], s->ic, FMA(-M[0], s->ib, a))};
#else
  const REAL eW0 = w[0] - r_cmd * E02, eW1 = w[1] - r_cmd * E12, eW2 = w[2] - r_cmd * E22;
  M[0] = -(eR0 * s->kR[0]) - eW0 * s->kW[0];
  M[1] = -(eR1 * s->kR[1]) - eW1 * s->kW[1];
  M[2] = -(eR2 * s->kR[2]) - eW2 * s->kW[2];
  const REAL a = thrust * s->ia, bx = M[0] * s->ib, by = M[1] * s->ib, cz = M[2] * s->ic;
  const REAL w2[4] = {a - by + cz, a + bx - cz, a + by + cz, a - bx - cz};
#endif
#if ORACLE_F32 /* the kernel's sqrt_pos is the correctly rounded sqrt for x >= 2^-102 (exhaustively verified, dql_device.hpp): a stopped rotor is commanded 1e-15 rad/s */
  /* round 4: both clamps in one v_med3 — the upper one, at omax^2, replaces the motor model's min(cmd, omax) (exact for a correctly rounded sqrt and
   * an omax whose square is a float); motor_and_body(pre_clipped) then takes the command as the reference */
  for (int i = 0; i < 4; ++i) cmd[i] = SQRT(clip3(w2[i], 1e-30f, s->omax * s->omax));
#else
  for (int i = 0; i < 4; ++i) cmd[i] = SQRT(w2[i] > R_(0.0) ? w2[i] : R_(0.0));
#endif
}

/* gazebo_motor_model.cpp:434-500 (forces from the CURRENT rotor speeds) + one semi-implicit Euler step of the body */
static inline void motor_and_body(const simc_t* s, env_t* e, const REAL R[9], const REAL cmd[4], int pre_clipped) {
  const REAL* om = e->om; const REAL* w = e->w; const REAL l = s->l, h = s->h;
  /* thrust k_f om_i^2 along body z at rotor i = (+l,0,h), (0,+l,h), (-l,0,h), (0,-l,h); drag torque -dir_i k_m T_i
   * (gazebo_motor_model.cpp:441-452, 476-482; allocation signs pkg/attitude_controller.py:98-104) */
#if ORACLE_F32
  /* round 4b: opposite rotors first — sums and differences of the speeds of arm x (0, 2) and arm y (1, 3) serve the thrust torques
   * (q1 - q3 = (om1 - om3)(om1 + om3)), the drag sums and the total alike; squares by fma: 17 operations instead of 21 (csrc/dql_device.hpp plant_step) */
  const REAL s02 = om[0] + om[2], s13 = om[1] + om[3], d02 = om[0] - om[2], d13 = om[1] - om[3];
  const REAL S = s02 + s13;
  const REAL qa_ = FMA(om[0], om[0], om[2] * om[2]), qb_ = FMA(om[1], om[1], om[3] * om[3]); /* q0 + q2, q1 + q3 */
  const REAL Fbz = s->kf * (qa_ + qb_);
  REAL tx = s->lkf * (d13 * s13), ty = -(s->lkf * (d02 * s02)), tz = s->kmkf * (qa_ - qb_);
#else
  const REAL q0 = om[0] * om[0], q1 = om[1] * om[1], q2 = om[2] * om[2], q3 = om[3] * om[3];
  const REAL Fbz = s->kf * ((q0 + q1) + (q2 + q3));
  REAL tx = s->lkf * (q1 - q3), ty = s->lkf * (q2 - q0), tz = s->kmkf * ((q0 - q1) + (q2 - q3));
#endif
  /* rotor drag -|om_i| c_d v_perp,i (gazebo_motor_model.cpp:458-469), v_perp,i = (v_body + w x r_i) in the rotor plane,
   * summed over the four rotors in closed form; rolling moment (:484-489) = (c_r / c_d) x the drag force */
  const REAL vbx = FMA(R[0], e->v[0], FMA(R[3], e->v[1], R[6] * e->v[2]));
  const REAL vby = FMA(R[1], e->v[0], FMA(R[4], e->v[1], R[7] * e->v[2]));
  const REAL uxc = FMA(w[1], h, vbx), uyc = FMA(-w[0], h, vby), wzl = w[2] * l;
#if !ORACLE_F32
  const REAL S = (om[0] + om[1]) + (om[2] + om[3]), d02 = om[0] - om[2], d13 = om[1] - om[3];
#endif
  const REAL Fbx = -(s->cd * FMA(S, uxc, -(wzl * d13)));
  const REAL Fby = -(s->cd * FMA(S, uyc, wzl * d02));
#if ORACLE_F32 /* the drag's yaw torque with -(l c_d) as one host constant */
  tz = FMA(s->nlcd, FMA(uyc, d02, FMA(wzl, S, -(uxc * d13))), tz);
  tx = FMA(-h, Fby, tx); ty = FMA(h, Fbx, ty);
#else
  const REAL tzd = -(s->cd * FMA(uyc, d02, FMA(wzl, S, -(uxc * d13))));
  tx = FMA(-h, Fby, tx); ty = FMA(h, Fbx, ty); tz = FMA(l, tzd, tz);
#endif
  tx = FMA(s->crd, Fbx, tx); ty = FMA(s->crd, Fby, ty);
  /* rotor speed filter (common.h:147-183), commanded speed clipped at max_rot_velocity (gazebo_motor_model.cpp:358-364) */
  for (int i = 0; i < 4; ++i) {
#if ORACLE_F32
    const REAL ref = pre_clipped ? cmd[i] : clip3(cmd[i], R_(0.0), s->omax);
#else
    (void)pre_clipped;
    const REAL ref = clip3(cmd[i], R_(0.0), s->omax);  /* cmd = sqrt(..) >= +0: min(cmd, omax) */
#endif
#if ORACLE_F32 /* om + (1 - a)(ref - om): the same filter, one operation less */
    const REAL d = ref - e->om[i];
    const REAL c = d > R_(0.0) ? s->oup : s->odn;
    e->om[i] = FMA(c, d, e->om[i]);
#else
    const REAL a = ref > e->om[i] ? s->aup : s->adn;
    e->om[i] = FMA(a, e->om[i], (R_(1.0) - a) * ref);
#endif
  }
  /* translation */
  const REAL Fwx = FMA(R[0], Fbx, FMA(R[1], Fby, R[2] * Fbz)), Fwy = FMA(R[3], Fbx, FMA(R[4], Fby, R[5] * Fbz)), Fwz = FMA(R[6], Fbx, FMA(R[7], Fby, R[8] * Fbz));
#if ORACLE_F32
  e->v[0] = FMA(s->dtm, Fwx, e->v[0]); e->v[1] = FMA(s->dtm, Fwy, e->v[1]); e->v[2] = FMA(s->dtm, Fwz, e->v[2]) - s->dtg;
#else
  const REAL ax = Fwx * s->inv_m, ay = Fwy * s->inv_m, az = Fwz * s->inv_m - s->g;
  e->v[0] = FMA(s->dt, ax, e->v[0]); e->v[1] = FMA(s->dt, ay, e->v[1]); e->v[2] = FMA(s->dt, az, e->v[2]);
#endif
  e->p[0] = FMA(s->dt, e->v[0], e->p[0]); e->p[1] = FMA(s->dt, e->v[1], e->p[1]); e->p[2] = FMA(s->dt, e->v[2], e->p[2]);
  /* rotation: I w' = tau - w x I w */
  const REAL Iw0 = s->I[0] * w[0], Iw1 = s->I[1] * w[1], Iw2 = s->I[2] * w[2];
  const REAL g0 = FMA(w[1], Iw2, -(w[2] * Iw1)), g1 = FMA(w[2], Iw0, -(w[0] * Iw2)), g2 = FMA(w[0], Iw1, -(w[1] * Iw0));
#if ORACLE_F32
  { const REAL w0 = w[0], w1 = w[1], w2_ = w[2];
    e->w[0] = FMA(s->dtI[0], tx - g0, w0); e->w[1] = FMA(s->dtI[1], ty - g1, w1);
    /* round 4: (w x I w)_z = (I_y - I_x) w_x w_y is exactly 0 for a vehicle with I_x = I_y (the reference's): not computed */
    if (s->I[0] == s->I[1]) e->w[2] = FMA(s->dtI[2], tz, w2_);
    else e->w[2] = FMA(s->dtI[2], tz - g2, w2_); }
#else
  e->w[0] = FMA(s->dt, (tx - g0) * s->inv_I[0], w[0]);
  e->w[1] = FMA(s->dt, (ty - g1) * s->inv_I[1], w[1]);
  e->w[2] = FMA(s->dt, (tz - g2) * s->inv_I[2], w[2]);
#endif
  const REAL qw = e->q[0], qx = e->q[1], qy = e->q[2], qz = e->q[3];
#if ORACLE_F32 /* round 4b: the body rates scaled by dt / 2 once, every component three fmas onto the old one: 15 operations instead of 17 */
  const REAL h0 = s->hdt * w[0], h1 = s->hdt * w[1], h2 = s->hdt * w[2];
  const REAL nw = FMA(-qx, h0, FMA(-qy, h1, FMA(-qz, h2, qw)));
  const REAL nx = FMA(qw, h0, FMA(qy, h2, FMA(-qz, h1, qx)));
  const REAL ny = FMA(qw, h1, FMA(qz, h0, FMA(-qx, h2, qy)));
  const REAL nz = FMA(qw, h2, FMA(qx, h1, FMA(-qy, h0, qz)));
#else
  const REAL hdt = R_(0.5) * s->dt;
  const REAL dw = -FMA(qx, w[0], FMA(qy, w[1], qz * w[2]));
  const REAL dxq = FMA(qw, w[0], FMA(qy, w[2], -(qz * w[1])));
  const REAL dyq = FMA(qw, w[1], FMA(qz, w[0], -(qx * w[2])));
  const REAL dzq = FMA(qw, w[2], FMA(qx, w[1], -(qy * w[0])));
  const REAL nw = FMA(hdt, dw, qw), nx = FMA(hdt, dxq, qx), ny = FMA(hdt, dyq, qy), nz = FMA(hdt, dzq, qz);
#endif
  /* renormalise with one Newton step of 1/sqrt(|q|^2) about 1: |q|^2 - 1 = O((dt |w|)^2), so the residual is O(dt^4) */
  const REAL inv = FMA(R_(-0.5), FMA(nw, nw, FMA(nx, nx, FMA(ny, ny, nz * nz))), R_(1.5));
  e->q[0] = nw * inv; e->q[1] = nx * inv; e->q[2] = ny * inv; e->q[3] = nz * inv;
}

/* pkg/moving_platform.py:87-127, phase = omega * t kept wrapped in [0, 2 pi) */
/* rec (float32 fused step, round 4; csrc/dql_device.hpp platform_update): sine / cosine evaluated at the period's first manager tick and carried to
 * its other ticks by the rotation through the constant phase step delta = omega mp_dt; NULL = evaluate at this tick */
typedef struct { REAL sn, cs, sd, cd; } platrec_t;
static inline void platform_update(const simc_t* s, env_t* e, platrec_t* rec, int first_in_period) {
  REAL sn, cs;
#if ORACLE_F32
  if (rec) {
    if (first_in_period) {
      det_sincos(e->mp_phase, &rec->sn, &rec->cs);
      const REAL d = e->mp_w * s->mp_dt;
      if (d > R_(0.25) || d < R_(-0.25)) det_sincos(d, &rec->sd, &rec->cd);
      else {
        const REAL z = d * d;
        rec->sd = d * FMA(z, FMA(z, R_(8.33333333333333322e-03), R_(-1.66666666666666657e-01)), R_(1.0));
        rec->cd = FMA(z, FMA(z, FMA(z, R_(-1.38888888888888894e-03), R_(4.16666666666666644e-02)), R_(-0.5)), R_(1.0));
      }
    }
    sn = rec->sn; cs = rec->cs;
    rec->sn = FMA(sn, rec->cd, cs * rec->sd);
    rec->cs = FMA(cs, rec->cd, -(sn * rec->sd));
  } else
#endif
  { (void)rec; (void)first_in_period; det_sincos(e->mp_phase, &sn, &cs); }
  if (s->traj == DQL_TRAJ_EIGHT) {
    e->mp_x = e->mp_r * cs; e->mp_y = e->mp_r * sn * cs;
    e->mp_u = -(e->mp_r * e->mp_w) * sn; e->mp_v = e->mp_r * e->mp_w * (cs * cs - sn * sn);
  } else {
    e->mp_x = e->mp_r * sn; e->mp_y = R_(0.0);
    e->mp_u = e->mp_r * e->mp_w * cs; e->mp_v = R_(0.0);
  }
  REAL ph = FMA(e->mp_w, s->mp_dt, e->mp_phase);
  if (ph >= R_(6.28318530717958623200e+00)) ph -= R_(6.28318530717958623200e+00);
  e->mp_phase = ph;
}

/* scripts/manager_node.py:192-214 + pkg/observation_utils.py:77-158: relative state in the yaw-only frame, PID
 * inputs, acceleration estimate; THEN the platform set-point for the next 10 ms (the observation uses the platform
 * state Gazebo reported before this tick's set_model_state) */
/* with_noise: apply this tick's noise draw.  The noise sits on the published p / v only, every tick overwrites the latched
 * Observation and the MDP reads the latch once per agent period, so inside the fused step only the period's LAST manager tick needs
 * its draw; the others are skipped (same values, bit for bit: tests/test_oracle_golden.py::test_lazy_noise_equals_eager_noise pins
 * it against drawing every tick, orc_set_eager_noise) */
static int g_eager_noise = 0;
EXPORT void ORC(set_eager_noise)(int on) { g_eager_noise = on; }
static inline void manager_tick(const simc_t* s, env_t* e, const REAL R[9], REAL cy, REAL sy, int64_t mgr_index,
                                uint32_t k0, uint32_t k1, uint32_t step_lo, uint32_t step_hi, uint32_t env_id, uint32_t mgr_in_step, int with_noise, platrec_t* rec) {
  const REAL dxw = e->mp_x - e->p[0], dyw = e->mp_y - e->p[1];
  const REAL dvx = e->mp_u - e->v[0], dvy = e->mp_v - e->v[1];
  const REAL rpx = FMA(cy, dxw, sy * dyw), rpy = FMA(cy, dyw, -(sy * dxw));
  const REAL rvx = FMA(cy, dvx, sy * dvy), rvy = FMA(cy, dvy, -(sy * dvx));
  /* PID plant states: scripts/manager_node.py:292-310 */
  e->vz.state = e->v[2];                       /* -(0 - v_z) */
  {
    /* yaw of q_drone * q_platform^-1 in the stability frame = yaw of Rz(-psi) R Rz(psi) */
    const REAL A00 = FMA(cy, R[0], sy * R[3]), A01 = FMA(cy, R[1], sy * R[4]);
    const REAL A10 = FMA(cy, R[3], -(sy * R[0])), A11 = FMA(cy, R[4], -(sy * R[1]));
    e->yaw.state = det_atan2(FMA(A10, cy, A11 * sy), FMA(A00, cy, A01 * sy));
  }
  REAL opx = rpx, opy = rpy, ovx = rvx, ovy = rvy;
  if (with_noise && (s->noise_p > R_(0.0) || s->noise_v > R_(0.0))) { /* pkg/observation_utils.py:127-128 */
    uint32_t r[4]; REAL n0, n1, n2, n3;
    philox4x32(step_lo, step_hi, env_id, STREAM_NOISE0 + mgr_in_step, k0, k1, r);
    box_muller(r[0], r[1], &n0, &n1); box_muller(r[2], r[3], &n2, &n3);
    opx = FMA(s->noise_p, n0, opx); opy = FMA(s->noise_p, n1, opy); ovx = FMA(s->noise_v, n2, ovx); ovy = FMA(s->noise_v, n3, ovy);
  }
  REAL ax_ = R_(0.0), ay_ = R_(0.0);
  if (!(e->flags & FL_ACC_INIT)) { /* pkg/observation_utils.py:137-143 */
    e->vf[0] = rvx; if (s->two_axis) e->vf[1] = rvy; e->flags |= FL_ACC_INIT;
  } else {
    REAL dt_;
    if (s->quirks & DQL_Q_FROZEN_ACC_REFERENCE) dt_ = (REAL)mgr_index * s->mgr_dt; /* time since the first sample; B19 */
    else dt_ = s->mgr_dt;
    if (dt_ <= R_(0.0)) dt_ = R_(0.01); /* pkg/filters.py:67-69 */
#if ORACLE_F32
    if (!(s->quirks & DQL_Q_FROZEN_ACC_REFERENCE)) { /* constant divisor: one multiplication (float32 tick) */
      ax_ = kalman1d(&e->kal_x[0], &e->kal_P[0], s->kal_q, s->kal_r, (rvx - e->vf[0]) * s->inv_mgr_dt);
      if (s->two_axis) ay_ = kalman1d(&e->kal_x[1], &e->kal_P[1], s->kal_q, s->kal_r, (rvy - e->vf[1]) * s->inv_mgr_dt);
    } else
#endif
    {
    ax_ = kalman1d(&e->kal_x[0], &e->kal_P[0], s->kal_q, s->kal_r, (rvx - e->vf[0]) / dt_);
    if (s->two_axis) ay_ = kalman1d(&e->kal_x[1], &e->kal_P[1], s->kal_q, s->kal_r, (rvy - e->vf[1]) / dt_); /* y estimator only flies in 2-axis configs */
    }
    if (!(s->quirks & DQL_Q_FROZEN_ACC_REFERENCE)) { e->vf[0] = rvx; if (s->two_axis) e->vf[1] = rvy; }
  }
  e->obs[0] = opx; e->obs[1] = opy; e->obs[2] = ovx; e->obs[3] = ovy; e->obs[4] = ax_; e->obs[5] = ay_;
  /* Observation.contact = the latched bumper flag at publish time (pkg/observation_utils.py:156) */
  if (e->flags & FL_CONTACT) e->flags |= FL_OBS_CONTACT; else e->flags &= ~FL_OBS_CONTACT;
  platform_update(s, e, rec, mgr_in_step == 0);
}

/* drone start coordinate along one axis from the random offset x0 and the platform coordinate (init_mode = cfg.init_uniform):
 * 0 / 1  TrainingLandingEnv.reset (pkg/landing_simulation_env.py:205-209): clip(x0 + mp, mp - p_max, mp + p_max)
 * 2      SimulationLandingEnv.reset (:339-343): clip(mp - x0, -p_max, p_max) — the offset is subtracted and the clip is absolute */
static inline REAL place_axis(int init_mode, REAL x0, REAL mp, REAL p_max) {
  if (init_mode == 2) return clip(mp - x0, -p_max, p_max);
  return clip(x0 + mp, mp - p_max, mp + p_max);
}

typedef struct { int64_t decisions, episodes, by_code[DQL_N_CHECK_CODES], reward_fx; } ostats_t;

/* One agent period for one env: pkg/trainer.py:191-212 body (guess, env.step, TD target) or, for an env whose
 * episode ended, pkg/landing_simulation_env.py:167-243 (reset: placement + one agent period + first state).
 * mode: 0 = eps-greedy + accumulate TD targets, 1 = greedy / no learning, 2 = external actions / no learning */
static void env_agent_period(const simc_t* s, const mdpc_t* m, env_t* e, const double* qa, const double* qb, int64_t* accum,
                             ostats_t* st, int mode, double eps, const uint8_t* ext_action, uint64_t seed, uint32_t env_id,
                             int64_t step_index, int64_t g0, int n_ticks, double gamma) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), step_lo = (uint32_t)step_index, step_hi = (uint32_t)((uint64_t)step_index >> 32);
  uint32_t r[4];
  philox4x32(step_lo, step_hi, env_id, STREAM_ACTION, k0, k1, r);
  const int is_reset = (e->flags & FL_DONE) != 0;
  const int prev_idx = e->idx[0], prev_idy = e->idx[1], two = s->two_axis;
  int action = 2, action_y = 2;
  uint32_t r2[4] = {0u, 0u, 0u, 0u};
  if (two) philox4x32(step_lo, step_hi, env_id, STREAM_ACTION + 1u, k0, k1, r2); /* y axis: same draws on its own stream */
  if (is_reset) {
    /* TrainingMdp.reset (pkg/mdp.py:562-569, 194-200) */
    e->step_count = 0; e->cur_check = 0; e->code = DQL_NON_TERMINAL; e->cum[0] = R_(0.0); e->cum[1] = R_(0.0);
    e->pitch_sp = R_(0.0); e->roll_sp = R_(0.0);
    if (!(s->quirks & DQL_Q_SHAPING_SURVIVES_RESET)) { for (int a = 0; a < 2; ++a) for (int k = 0; k < 3; ++k) e->shp[a][k] = R_(0.0); }
    /* placement (pkg/landing_simulation_env.py:181-218) */
    REAL x0;
    if (s->working == 0 && !s->init_uniform) { REAL n0, n1; box_muller(r[2], r[3], &n0, &n1); x0 = s->init_sigma * n0; }
    else x0 = FMA(R_(2.0) * u24(r[2]), s->p_max, -s->p_max);
    e->p[0] = place_axis(s->init_uniform, x0, e->mp_x, s->p_max);
    e->p[1] = R_(0.0); e->p[2] = s->z_init;
    if (two) { /* the reference multiplies its y offset by 0 (B16); the 2-axis configs fly it */
      REAL y0;
      if (s->working == 0 && !s->init_uniform) { REAL n0, n1; box_muller(r2[2], r2[3], &n0, &n1); y0 = s->init_sigma * n0; }
      else y0 = FMA(R_(2.0) * u24(r2[2]), s->p_max, -s->p_max);
      e->p[1] = place_axis(s->init_uniform, y0, e->mp_y, s->p_max);
    }
    e->v[0] = e->v[1] = e->v[2] = R_(0.0); e->w[0] = e->w[1] = e->w[2] = R_(0.0);
    e->q[0] = R_(1.0); e->q[1] = e->q[2] = e->q[3] = R_(0.0);
    e->flags &= ~(FL_DONE | FL_CONTACT | FL_OBS_CONTACT); /* scripts/manager_node.py:330 */
    e->flags |= FL_WAS_RESET;
  } else {
    e->flags &= ~FL_WAS_RESET;
    if (mode == 2) { action = ext_action[0] & 3; action_y = two ? (ext_action[0] >> 2) & 3 : 2; }
    else {
      /* guess (pkg/double_q_learning.py:110-117) */
      const int greedy = agent_predict(qa, qb, prev_idx);
      const int explore = (mode == 0) && ((double)u24(r[0]) < eps);
      action = explore ? (int)(((uint64_t)r[1] * 3u) >> 32) : greedy;
      if (two) {
        const int greedy_y = agent_predict(qa, qb, prev_idy);
        const int explore_y = (mode == 0) && ((double)u24(r2[0]) < eps);
        action_y = explore_y ? (int)(((uint64_t)r2[1] * 3u) >> 32) : greedy_y;
      }
    }
    e->pitch_sp = continuous_action(m, e->pitch_sp, action);
    /* y-axis angle theta_y = -roll: a positive theta_y accelerates the drone towards +y, as a positive pitch does towards +x */
    if (two) e->roll_sp = -continuous_action(m, -e->roll_sp, action_y);
  }
  e->action = action | (two ? action_y << 2 : 0);
  /* B = Rx(roll_sp) Ry(pitch_sp) (pkg/attitude_controller.py:138-140) */
  REAL sp_, cp_, sr_, cr_, B[9];
  det_sincos(e->pitch_sp, &sp_, &cp_); det_sincos(e->roll_sp, &sr_, &cr_);
  B[0] = cp_; B[1] = R_(0.0); B[2] = sp_;
  B[3] = sr_ * sp_; B[4] = cr_; B[5] = -(sr_ * cp_);
  B[6] = -(cr_ * sp_); B[7] = sr_; B[8] = cr_ * cp_;
  REAL R[9], cy, sy, ct, rn;
  uint32_t mgr_in_step = 0;
  platrec_t prec = {R_(0.0), R_(0.0), R_(0.0), R_(0.0)};
  const int phase0 = (int)(g0 % s->div), first_mgr = phase0 ? s->div - phase0 : 0;
  const uint32_t last_mgr = first_mgr < n_ticks ? (uint32_t)((n_ticks - 1 - first_mgr) / s->div) : 0u; /* the period's last manager tick */
  for (int i = 0; i < n_ticks; ++i) {
    const int64_t g = g0 + i;
    quat_to_R(e->q, R); yaw_cs4(R, &cy, &sy, &ct, &rn);
    if (g % s->div == 0) { manager_tick(s, e, R, cy, sy, g / s->div, k0, k1, step_lo, step_hi, env_id, mgr_in_step, g_eager_noise || mgr_in_step == last_mgr, &prec); ++mgr_in_step; }
    const REAL thrust = pid_output(&s->pvz, &s->bw, &e->vz, s->dt);
    const REAL r_cmd = pid_output(&s->pyaw, &s->bw, &e->yaw, s->dt);
    REAL cmd[4], M[3];
    attitude(s, R, e->w, B, cy, sy, ct, rn, r_cmd, thrust, cmd, M, !s->two_axis);
    motor_and_body(s, e, R, cmd, 1);
    e->mp_x = FMA(e->mp_u, s->dt, e->mp_x); e->mp_y = FMA(e->mp_v, s->dt, e->mp_y);
#if ORACLE_F32 /* round 4b: altitude test against the host's mp_top + bottom */
    if (e->p[2] <= s->low_z && FABS(e->p[0] - e->mp_x) <= s->mp_hx && FABS(e->p[1] - e->mp_y) <= s->mp_hy) e->flags |= FL_CONTACT;
#else
    if (e->p[2] - s->bottom <= s->mp_top && FABS(e->p[0] - e->mp_x) <= s->mp_hx && FABS(e->p[1] - e->mp_y) <= s->mp_hy) e->flags |= FL_CONTACT;
#endif
  }
  /* euler_from_quaternion, axes sxyz (pkg/landing_simulation_env.py:259-267) */
  quat_to_R(e->q, R);
  /* discrete_state (pkg/mdp.py:257-333) on the latest latched Observation + fresh pitch / altitude */
  int idx, idy = -1;
#if ORACLE_F32 /* round 5: the angle bins straight from the rotation matrix (pitch = atan2(-R20, sqrt(R00^2 + R10^2)), roll = atan2(R21, R22)) */
  idx = discretise_bin(m, e->obs[0], e->obs[2], e->obs[4], R_(0.0), angle_bin_from_tangent(m, -R[6], FMA(R[0], R[0], R[3] * R[3]), 1));
  if (two) idy = discretise_bin(m, e->obs[1], e->obs[3], e->obs[5], R_(0.0), angle_bin_from_tangent(m, -R[7], R[8] * R[8], R[8] > R_(0.0)));
#else
  const REAL cyy = SQRT(FMA(R[0], R[0], R[3] * R[3]));
  const REAL pitch = det_atan2(-R[6], cyy);
  idx = discretise(m, e->obs[0], e->obs[2], e->obs[4], pitch);
  if (two) { const REAL roll = det_atan2(R[7], R[8]); idy = discretise(m, e->obs[1], e->obs[3], e->obs[5], -roll); }
#endif
  if (idx < 0) idx = 0; /* only reachable through NaN; the reference raises ValueError there */
  e->idx[0] = idx;
  if (two) {
    if (idy < 0) idy = 0;
    e->idx[1] = idy;
  }
  e->reward = R_(0.0);
  if (is_reset) return;
  const int contact = (e->flags & FL_OBS_CONTACT) != 0;
  e->code = mdp_check(m, &e->step_count, &e->cur_check, e->code, prev_idx, idx, contact, e->obs[0], e->obs[1], e->p[2], two, prev_idy, idy);
  REAL rew = mdp_reward(m, e->shp[0], &e->cum[0], e->code, idx, e->obs[0], e->obs[2], e->pitch_sp);
  REAL rew_y = R_(0.0);
  if (two) rew_y = mdp_reward(m, e->shp[1], &e->cum[1], e->code, idy, e->obs[1], e->obs[3], -e->roll_sp);
  e->reward = two ? rew + rew_y : rew;
  const int done = e->code <= DQL_TERMINAL_TIMEOUT;
  if (done) e->flags |= FL_DONE;
  st->decisions += 1;
  st->reward_fx += fx_round((double)rew * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (two) st->reward_fx += fx_round((double)rew_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
  if (done) { st->episodes += 1; st->by_code[e->code] += 1; }
  if (mode == 0) {
    /* TD target of _update_q_table (pkg/double_q_learning.py:136-145), accumulated in fixed point: accum = [4][N_CELLS] =
     * {target sums, visits} of Q_table_a, then of Q_table_b.  Reference (B1/B2, DQL_Q_UPDATE_TABLE_A_ONLY): always table a,
     * valued by itself.  Without the quirk, Double Q-learning as the paper has it: a fair coin (bit 31 of the third word of
     * the period's action stream; that word places the vehicle in reset periods and is free in all others) picks the table
     * to update, and the OTHER table values the picked table's greedy action at s'. */
    const int dbl = !(s->quirks & DQL_Q_UPDATE_TABLE_A_ONLY);
    {
      const int sel_b = dbl && (r[2] >> 31);
      const double* qsel = (sel_b ? qb : qa) + idx * 3;
      const double* qval = dbl ? (sel_b ? qa : qb) + idx * 3 : qsel;
      const double boot = qval[argmax3(qsel[0], qsel[1], qsel[2])];
      int mask = 1;
      if (s->quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask = idx_pos(prev_idx) != idx_pos(idx);
      else mask = !done;
      const double target = (double)rew + (gamma * boot) * (double)mask;
      const int cell = prev_idx * 3 + action;
      int64_t* acc = accum + (sel_b ? 2 * DQL_N_CELLS : 0);
      acc[cell] += fx_round(target * (double)(1ll << DQL_TARGET_FRAC_BITS));
      acc[DQL_N_CELLS + cell] += 1;
    }
    if (two) { /* the y transition updates the same shared tables (scripts/simulation.py:15-16 loads one table pair for both axes) */
      const int sel_b = dbl && (r2[2] >> 31);
      const double* qsel = (sel_b ? qb : qa) + idy * 3;
      const double* qval = dbl ? (sel_b ? qa : qb) + idy * 3 : qsel;
      const double boot_y = qval[argmax3(qsel[0], qsel[1], qsel[2])];
      int mask_y;
      if (s->quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) mask_y = idx_pos(prev_idy) != idx_pos(idy);
      else mask_y = !done;
      const double target_y = (double)rew_y + (gamma * boot_y) * (double)mask_y;
      const int cell_y = prev_idy * 3 + action_y;
      int64_t* acc = accum + (sel_b ? 2 * DQL_N_CELLS : 0);
      acc[cell_y] += fx_round(target_y * (double)(1ll << DQL_TARGET_FRAC_BITS));
      acc[DQL_N_CELLS + cell_y] += 1;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * exported entry points (ctypes)
 * ---------------------------------------------------------------------------------------------- */
EXPORT int ORC(env_size)(void) { return (int)sizeof(env_t); }
EXPORT int ORC(n_fields)(int is_int) { return is_int ? NF_INT : NF_REAL; }
EXPORT const char* ORC(field_name)(int i, int is_int) { return is_int ? k_int_names[i] : k_real_names[i]; }

/* creation: hover rotor speeds, thrust integral at hover, platform phase (and per-env platform) from the init stream */
EXPORT void ORC(init_envs)(const dql_config* c, void* envs_, int64_t n, uint64_t seed, int64_t env_id_offset) {
  env_t* envs = (env_t*)envs_;
  simc_t s; simc_init(&s, c);
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const REAL hover = SQRT((REAL)(c->mass * c->gravity / (4.0 * c->k_f)));
  for (int64_t i = 0; i < n; ++i) {
    env_t* e = &envs[i];
    memset(e, 0, sizeof(*e));
    uint32_t r[4];
    philox4x32(0u, 0u, (uint32_t)(env_id_offset + i), STREAM_INIT, k0, k1, r);
    e->q[0] = R_(1.0);
    e->p[2] = s.z_init;
    for (int k = 0; k < 4; ++k) e->om[k] = hover;
    e->vz.integ = (REAL)(c->mass * c->gravity / c->pid_vz[1]);
    e->kal_P[0] = e->kal_P[1] = R_(1.0); /* pkg/filters.py:15 */
    e->mp_r = s.mp_r; e->mp_w = s.mp_w;
    if (s.per_env_platform && s.traj == DQL_TRAJ_RPM) {
      e->mp_r = FMA(u24(r[1]), s.mp_r_hi - s.mp_r_lo, s.mp_r_lo);
      const REAL tx = FMA(u24(r[2]), s.mp_t_hi - s.mp_t_lo, s.mp_t_lo);
      e->mp_w = tx / e->mp_r;
    }
    e->mp_phase = R_(6.28318530717958623200e+00) * u24(r[0]);
    /* platform state at t = 0 without advancing the phase */
    REAL sn, cs; det_sincos(e->mp_phase, &sn, &cs);
    if (s.traj == DQL_TRAJ_EIGHT) { e->mp_x = e->mp_r * cs; e->mp_y = e->mp_r * sn * cs; e->mp_u = -(e->mp_r * e->mp_w) * sn; e->mp_v = e->mp_r * e->mp_w * (cs * cs - sn * sn); }
    else { e->mp_x = e->mp_r * sn; e->mp_u = e->mp_r * e->mp_w * cs; }
    e->code = DQL_NON_TERMINAL;
    e->idx[0] = e->idx[1] = -1;
    e->flags = FL_DONE; /* every env enters through reset */
    e->action = 2;
  }
}

/* n_threads > 1: envs are split over OpenMP threads (cpu_baseline leg of bench.py); every thread adds into its own
 * accumulators, merged with integer sums, so the result does not depend on the thread count */
EXPORT void ORC(agent_periods)(const dql_config* c, void* envs_, int64_t n, const double* qa, const double* qb, int64_t* accum,
                               int64_t* stats_out /* [2 + 9 + 1] */, int mode, double eps, const uint8_t* ext_actions, uint64_t seed,
                               int64_t env_id_offset, int64_t step_index, int64_t g0, int n_ticks, int n_threads) {
  env_t* envs = (env_t*)envs_;
  simc_t s; simc_init(&s, c);
  mdpc_t m; mdpc_init(&m, c);
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads) if (n_threads > 1)
  {
    ostats_t st; memset(&st, 0, sizeof(st));
    int64_t* acc = n_threads > 1 ? (int64_t*)calloc(4 * DQL_N_CELLS, sizeof(int64_t)) : accum;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i)
      env_agent_period(&s, &m, &envs[i], qa, qb, acc, &st, mode, eps, ext_actions ? ext_actions + i : 0, seed,
                       (uint32_t)(env_id_offset + i), step_index, g0, n_ticks, c->gamma);
#pragma omp critical
    {
      if (n_threads > 1) { for (int k = 0; k < 4 * DQL_N_CELLS; ++k) accum[k] += acc[k]; free(acc); }
      stats_out[0] += st.decisions; stats_out[1] += st.episodes;
      for (int k = 0; k < DQL_N_CHECK_CODES; ++k) stats_out[2 + k] += st.by_code[k];
      stats_out[11] += st.reward_fx;
    }
  }
}

EXPORT void ORC(get_fields)(const void* envs_, int64_t n, double* reals /*[NF_REAL][n]*/, int32_t* ints /*[NF_INT][n]*/) {
  const env_t* envs = (const env_t*)envs_;
  double f[NF_REAL]; int32_t g[NF_INT];
  for (int64_t i = 0; i < n; ++i) {
    env_to_fields(&envs[i], f, g);
    for (int k = 0; k < NF_REAL; ++k) reals[(int64_t)k * n + i] = f[k];
    for (int k = 0; k < NF_INT; ++k) ints[(int64_t)k * n + i] = g[k];
  }
}
EXPORT void ORC(set_fields)(void* envs_, int64_t n, const double* reals, const int32_t* ints) {
  env_t* envs = (env_t*)envs_;
  double f[NF_REAL]; int32_t g[NF_INT];
  for (int64_t i = 0; i < n; ++i) {
    for (int k = 0; k < NF_REAL; ++k) f[k] = reals[(int64_t)k * n + i];
    for (int k = 0; k < NF_INT; ++k) g[k] = ints[(int64_t)k * n + i];
    fields_to_env(&envs[i], f, g);
  }
}

/* ---- stand-alone pieces, checked one by one against the golden vectors ---- */
EXPORT void ORC(discretise)(const dql_config* c, const double* p, const double* v, const double* a, const double* ang, int64_t n, int32_t* out) {
  mdpc_t m; mdpc_init(&m, c);
  for (int64_t i = 0; i < n; ++i) out[i] = discretise(&m, (REAL)p[i], (REAL)v[i], (REAL)a[i], (REAL)ang[i]);
}
/* one MDP transition per element; mdp_state double[8][n] = pitch_sp, shp_p, shp_v, shp_a, cumulative, step_count, cur_check, code */
EXPORT void ORC(mdp_transition)(const dql_config* c, int64_t n, const uint8_t* action, const double* obs /*[7][n]*/, double* ms,
                                const int32_t* prev_idx, int32_t* idx_out, double* reward_out, uint8_t* done_out) {
  mdpc_t m; mdpc_init(&m, c);
  for (int64_t i = 0; i < n; ++i) {
    REAL sp = (REAL)ms[0 * n + i], shp[3] = {(REAL)ms[1 * n + i], (REAL)ms[2 * n + i], (REAL)ms[3 * n + i]}, cum = (REAL)ms[4 * n + i];
    int step_count = (int)ms[5 * n + i], cur_check = (int)ms[6 * n + i], code = (int)ms[7 * n + i];
    sp = continuous_action(&m, sp, action[i]);
    const REAL px = (REAL)obs[0 * n + i], py = (REAL)obs[1 * n + i], vx = (REAL)obs[2 * n + i], ax = (REAL)obs[3 * n + i];
    const REAL pitch = (REAL)obs[4 * n + i], z = (REAL)obs[5 * n + i]; const int contact = obs[6 * n + i] != 0.0;
    const int idx = discretise(&m, px, vx, ax, pitch);
    idx_out[i] = idx;
    code = mdp_check(&m, &step_count, &cur_check, code, prev_idx[i], idx, contact, px, py, z, 0, -1, -1);
    const REAL rew = mdp_reward(&m, shp, &cum, code, idx, px, vx, sp);
    reward_out[i] = rew; done_out[i] = code <= DQL_TERMINAL_TIMEOUT;
    ms[0 * n + i] = sp; ms[1 * n + i] = shp[0]; ms[2 * n + i] = shp[1]; ms[3 * n + i] = shp[2]; ms[4 * n + i] = cum;
    ms[5 * n + i] = step_count; ms[6 * n + i] = cur_check; ms[7 * n + i] = code;
  }
}
EXPORT void ORC(butterworth_run)(double c, const double* x, int64_t n, double* y) {
  bwc_t b; bwc_init(&b, c);
  REAL x1 = 0, x2 = 0, y1 = 0, y2 = 0, y3 = 0;
  for (int64_t i = 0; i < n; ++i) y[i] = butterworth(&b, (REAL)x[i], &x1, &x2, &y1, &y2, &y3);
}
/* KalmanFilter3D.filter over a velocity series (pkg/filters.py:53-80); dt_le0[i] marks the dt <= 0 branch */
EXPORT void ORC(kalman_run)(double q, double sd, const double* vel /*[n][3]*/, const uint8_t* dt_le0, int64_t n, double* acc /*[n-1][3]*/) {
  REAL x[3] = {0, 0, 0}, P[3] = {1, 1, 1};
  const REAL Q = (REAL)q, Rm = (REAL)(sd * sd);
  for (int64_t i = 1; i < n; ++i) {
    REAL dt_ = dt_le0[i] ? R_(0.0) : (REAL)(0.01 * (double)i) - (REAL)(0.01 * (double)(i - 1));
    if (dt_ <= R_(0.0)) dt_ = R_(0.01);
    for (int k = 0; k < 3; ++k) acc[(i - 1) * 3 + k] = kalman1d(&x[k], &P[k], Q, Rm, ((REAL)vel[i * 3 + k] - (REAL)vel[(i - 1) * 3 + k]) / dt_);
  }
}
/* PID.output replay: params = Kp Ki Kd lo hi windup setpoint; state sampled every 5th tick; tick times 0.002 (i+1) */
EXPORT void ORC(pid_run)(const double* params, double bw_c, const double* state, int64_t n, double* effort, double* integral) {
  pidc_t c = {(REAL)params[0], (REAL)params[1], (REAL)params[2], (REAL)params[3], (REAL)params[4], (REAL)params[5], (REAL)params[6]};
  bwc_t b; bwc_init(&b, bw_c);
  pid_t_ s; memset(&s, 0, sizeof(s));
  REAL prev_t = R_(0.0);
  for (int64_t i = 0; i < n; ++i) {
    const REAL t = (REAL)(0.002 * (double)(i + 1));
    if (i % 5 == 0) s.state = (REAL)state[i];
    effort[i] = pid_output(&c, &b, &s, t - prev_t);
    integral[i] = s.integ;
    prev_t = t;
  }
}
/* attitude law for n samples: quat (x,y,z,w) as in ROS, omega body, cmd = roll pitch yaw_rate thrust */
EXPORT void ORC(attitude_run)(const dql_config* c, const double* quat_xyzw, const double* omega, const double* cmd, int64_t n,
                              double* moment, double* rotor) {
  simc_t s; simc_init(&s, c);
  for (int64_t i = 0; i < n; ++i) {
    REAL q[4] = {(REAL)quat_xyzw[i * 4 + 3], (REAL)quat_xyzw[i * 4 + 0], (REAL)quat_xyzw[i * 4 + 1], (REAL)quat_xyzw[i * 4 + 2]};
    REAL w[3] = {(REAL)omega[i * 3], (REAL)omega[i * 3 + 1], (REAL)omega[i * 3 + 2]};
    REAL R[9], cy, sy, ct, rn, sp_, cp_, sr_, cr_, B[9], out[4], M[3];
    quat_to_R(q, R); yaw_cs4(R, &cy, &sy, &ct, &rn);
    det_sincos((REAL)cmd[i * 4 + 1], &sp_, &cp_); det_sincos((REAL)cmd[i * 4 + 0], &sr_, &cr_);
    B[0] = cp_; B[1] = R_(0.0); B[2] = sp_; B[3] = sr_ * sp_; B[4] = cr_; B[5] = -(sr_ * cp_); B[6] = -(cr_ * sp_); B[7] = sr_; B[8] = cr_ * cp_;
    attitude(&s, R, w, B, cy, sy, ct, rn, (REAL)cmd[i * 4 + 2], (REAL)cmd[i * 4 + 3], out, M, 0);
    for (int k = 0; k < 3; ++k) moment[i * 3 + k] = M[k];
    for (int k = 0; k < 4; ++k) rotor[i * 4 + k] = out[k];
  }
}
/* the same with the x-axis closed form of the float32 attitude law (roll command exactly 0; csrc/dql_device.hpp attitude(), xonly) */
EXPORT void ORC(attitude_run_x)(const dql_config* c, const double* quat_xyzw, const double* omega, const double* cmd, int64_t n, int xonly, double* rotor) {
  simc_t s; simc_init(&s, c);
  for (int64_t i = 0; i < n; ++i) {
    REAL q[4] = {(REAL)quat_xyzw[i * 4 + 3], (REAL)quat_xyzw[i * 4 + 0], (REAL)quat_xyzw[i * 4 + 1], (REAL)quat_xyzw[i * 4 + 2]};
    REAL w[3] = {(REAL)omega[i * 3], (REAL)omega[i * 3 + 1], (REAL)omega[i * 3 + 2]};
    REAL R[9], cy, sy, ct, rn, sp_, cp_, sr_, cr_, B[9], out[4], M[3];
    quat_to_R(q, R); yaw_cs4(R, &cy, &sy, &ct, &rn);
    det_sincos((REAL)cmd[i * 4 + 1], &sp_, &cp_); det_sincos((REAL)cmd[i * 4 + 0], &sr_, &cr_);
    B[0] = cp_; B[1] = R_(0.0); B[2] = sp_; B[3] = sr_ * sp_; B[4] = cr_; B[5] = -(sr_ * cp_); B[6] = -(cr_ * sp_); B[7] = sr_; B[8] = cr_ * cp_;
    attitude(&s, R, w, B, cy, sy, ct, rn, (REAL)cmd[i * 4 + 2], (REAL)cmd[i * 4 + 3], out, M, xonly);
    for (int k = 0; k < 4; ++k) rotor[i * 4 + k] = out[k];
  }
}
/* platform replay with the sine / cosine carried as the fused float32 step carries them inside an agent period: evaluated at every carry-th tick,
 * rotated through the constant phase step in between (carry = 0: evaluated at every tick) */
EXPORT void ORC(platform_run_carry)(const dql_config* c, int64_t n, int carry, double* out) {
  simc_t s; simc_init(&s, c);
  env_t e; memset(&e, 0, sizeof(e));
  e.mp_r = s.mp_r; e.mp_w = s.mp_w;
  platrec_t rec; memset(&rec, 0, sizeof(rec));
  for (int64_t i = 0; i < n; ++i) {
    if (carry > 0) platform_update(&s, &e, &rec, i % carry == 0);
    else platform_update(&s, &e, NULL, 1);
    out[i * 4] = e.mp_x; out[i * 4 + 1] = e.mp_y; out[i * 4 + 2] = e.mp_u; out[i * 4 + 3] = e.mp_v;
  }
}
/* platform trajectory replay from phase 0: out[n][4] = x, y, u, v at successive 100 Hz ticks */
EXPORT void ORC(platform_run)(const dql_config* c, int64_t n, double* out) {
  simc_t s; simc_init(&s, c);
  env_t e; memset(&e, 0, sizeof(e));
  e.mp_r = s.mp_r; e.mp_w = s.mp_w;
  for (int64_t i = 0; i < n; ++i) { platform_update(&s, &e, NULL, 1); out[i * 4] = e.mp_x; out[i * 4 + 1] = e.mp_y; out[i * 4 + 2] = e.mp_u; out[i * 4 + 3] = e.mp_v; }
}
/* the 100 Hz manager tick (scripts/manager_node.py:192-214,292-310 + pkg/observation_utils.py:77-158) replayed over a scripted
 * series: per tick in[14] = drone p(3), v(3), quaternion w x y z (world frame), platform x y u v as Gazebo reports them at this
 * tick; contact[t] = bumper flag.  out[11] = obs p_x p_y v_x v_y a_x a_y, v_z plant state, yaw plant state, then the platform
 * set-point x y u v published by this tick.  The acceleration estimator runs on both axes (two_axis forced on). */
EXPORT void ORC(manager_run)(const dql_config* c, int64_t n_ticks, const double* in /*[n_ticks][14]*/, const uint8_t* contact, uint64_t seed,
                             double* out /*[n_ticks][12]*/) {
  simc_t s; simc_init(&s, c);
  s.two_axis = 1;
  env_t e; memset(&e, 0, sizeof(e));
  e.kal_P[0] = e.kal_P[1] = R_(1.0);
  e.mp_r = s.mp_r; e.mp_w = s.mp_w;
  for (int64_t t = 0; t < n_ticks; ++t) {
    const double* r = in + t * 14;
    for (int k = 0; k < 3; ++k) { e.p[k] = (REAL)r[k]; e.v[k] = (REAL)r[3 + k]; }
    for (int k = 0; k < 4; ++k) e.q[k] = (REAL)r[6 + k];
    e.mp_x = (REAL)r[10]; e.mp_y = (REAL)r[11]; e.mp_u = (REAL)r[12]; e.mp_v = (REAL)r[13];
    if (contact[t]) e.flags |= FL_CONTACT;
    REAL R[9], cy, sy;
    quat_to_R(e.q, R); yaw_cs(R, &cy, &sy);
    manager_tick(&s, &e, R, cy, sy, t, (uint32_t)seed, (uint32_t)(seed >> 32), 0u, 0u, 0u, (uint32_t)t, 1, NULL);
    double* o = out + t * 12;
    o[0] = e.obs[0]; o[1] = e.obs[1]; o[2] = e.obs[2]; o[3] = e.obs[3]; o[4] = e.obs[4]; o[5] = e.obs[5];
    o[6] = e.vz.state; o[7] = e.yaw.state; o[8] = e.mp_x; o[9] = e.mp_y; o[10] = e.mp_u; o[11] = e.mp_v;
  }
}
/* the plant alone, open loop (include/dql.h dql_plant_run): per 500 Hz tick the rotor forces of the current rotor speeds + one
 * semi-implicit Euler step + rotor filter (motor_and_body), then platform extrapolation + contact latch.  init[21] = p v q(wxyz) w
 * om platform x y u v; cmd[n_ticks][4]; out[n_ticks][20] = p v q w om platform x y contact */
EXPORT void ORC(plant_run)(const dql_config* c, int64_t n_ticks, const double* init, const double* cmd_in, double* out) {
  simc_t s; simc_init(&s, c);
  env_t e; memset(&e, 0, sizeof(e));
  for (int k = 0; k < 3; ++k) { e.p[k] = (REAL)init[k]; e.v[k] = (REAL)init[3 + k]; e.w[k] = (REAL)init[10 + k]; }
  for (int k = 0; k < 4; ++k) { e.q[k] = (REAL)init[6 + k]; e.om[k] = (REAL)init[13 + k]; }
  e.mp_x = (REAL)init[17]; e.mp_y = (REAL)init[18]; e.mp_u = (REAL)init[19]; e.mp_v = (REAL)init[20];
  for (int64_t t = 0; t < n_ticks; ++t) {
    const REAL cmd[4] = {(REAL)cmd_in[t * 4], (REAL)cmd_in[t * 4 + 1], (REAL)cmd_in[t * 4 + 2], (REAL)cmd_in[t * 4 + 3]};
    REAL R[9];
    quat_to_R(e.q, R);
    motor_and_body(&s, &e, R, cmd, 0);
    e.mp_x = FMA(e.mp_u, s.dt, e.mp_x); e.mp_y = FMA(e.mp_v, s.dt, e.mp_y);
#if ORACLE_F32
    if (e.p[2] <= s.low_z && FABS(e.p[0] - e.mp_x) <= s.mp_hx && FABS(e.p[1] - e.mp_y) <= s.mp_hy) e.flags |= FL_CONTACT;
#else
    if (e.p[2] - s.bottom <= s.mp_top && FABS(e.p[0] - e.mp_x) <= s.mp_hx && FABS(e.p[1] - e.mp_y) <= s.mp_hy) e.flags |= FL_CONTACT;
#endif
    double* o = out + t * 20;
    for (int k = 0; k < 3; ++k) { o[k] = e.p[k]; o[3 + k] = e.v[k]; o[10 + k] = e.w[k]; }
    for (int k = 0; k < 4; ++k) { o[6 + k] = e.q[k]; o[13 + k] = e.om[k]; }
    o[17] = e.mp_x; o[18] = e.mp_y; o[19] = (e.flags & FL_CONTACT) ? 1.0 : 0.0;
  }
}
EXPORT void ORC(place)(const dql_config* c, const double* x0, const double* mp, int64_t n, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (double)place_axis(c->init_uniform, (REAL)x0[i], (REAL)mp[i], (REAL)c->p_max);
}
EXPORT void ORC(det_math)(const double* x, const double* y, int64_t n, double* s, double* c, double* at2, double* lg) {
  for (int64_t i = 0; i < n; ++i) {
    REAL ss, cc; det_sincos((REAL)x[i], &ss, &cc); s[i] = ss; c[i] = cc;
    at2[i] = det_atan2((REAL)y[i], (REAL)x[i]);
    lg[i] = det_log((REAL)(FABS((REAL)x[i]) > R_(1e-30) ? FABS((REAL)x[i]) : R_(1.0)));
  }
}

#if !ORACLE_F32
/* ---- precision-independent pieces, emitted once ---- */
EXPORT void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) { philox4x32(c0, c1, c2, c3, k0, k1, out); }
EXPORT void orc_agent_predict(const double* qa, const double* qb, const int32_t* idx, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)agent_predict(qa, qb, idx[i]);
}
/* DoubleQLearningAgent.update replayed in order (pkg/double_q_learning.py:91-146).  Reference quirks: B1/B2 (always table a,
 * valued by itself, DQL_Q_UPDATE_TABLE_A_ONLY) and B3 (bootstrap only when the position bin changed,
 * DQL_Q_BOOTSTRAP_ON_POS_CHANGE).  With B1/B2 cleared: Double Q-learning, coin[i] picks the table to update (the draw the
 * reference makes and ignores, :101), the other table values the picked table's greedy action; with B3 cleared the mask is
 * !done[i]. */
EXPORT void orc_agent_update(double* qa, double* qb, double* count, const int32_t* sa, const int32_t* ns, const double* alpha, double gamma,
                             const double* reward, int64_t n, uint32_t quirks, const uint8_t* coin, const uint8_t* done) {
  const int dbl = !(quirks & DQL_Q_UPDATE_TABLE_A_ONLY);
  for (int64_t i = 0; i < n; ++i) {
    count[sa[i]] += 1;
    const int sel_b = dbl && coin[i] != 0;
    double* qsel = sel_b ? qb : qa;
    const double* qval = dbl ? (sel_b ? qa : qb) : qa;
    const double* qn = qsel + ns[i] * 3;
    const double best = qval[ns[i] * 3 + argmax3(qn[0], qn[1], qn[2])];
    const int mask = (quirks & DQL_Q_BOOTSTRAP_ON_POS_CHANGE) ? (((sa[i] / 3) / 63) % 3 != (ns[i] / 63) % 3) : !done[i];
    const double loss = alpha[i] * (reward[i] + (gamma * best) * (double)mask - qsel[sa[i]]);
    qsel[sa[i]] += loss;
  }
}
/* transfer_learning (pkg/double_q_learning.py:77-89), k = 0 wraps to the last level (B6) */
EXPORT void orc_transfer(double* qa, double* qb, int k, double ratio, int n_levels) {
  const int src = (k - 1 + n_levels) % n_levels;
  for (int i = 0; i < DQL_CELLS_PER_LEVEL; ++i) {
    qa[k * DQL_CELLS_PER_LEVEL + i] = qa[src * DQL_CELLS_PER_LEVEL + i] * ratio;
    qb[k * DQL_CELLS_PER_LEVEL + i] = qb[src * DQL_CELLS_PER_LEVEL + i] * ratio;
  }
}
/* batched table update: for every cell visited m times with mean target tbar,
 *   Q <- tbar + (Q - tbar) * prod_{j<m} (1 - alpha(count + j)),  count += m
 * (m = 1 is the reference's Q += alpha (target - Q)); alpha(c) = alpha_tab[c] for c < n_tab, alpha_min beyond.
 * accum = [4][N_CELLS]: Q_table_a's {sums, visits}, then Q_table_b's; one shared visit counter (state_action_counter,
 * pkg/double_q_learning.py:100): table a's visits of a launch take the learning rates alpha(c) .. alpha(c + m_a - 1),
 * table b's the next m_b. */
/* per_step: one learning-rate step per launch the accumulators cover (n_launch = 1 for a launch's own fold, the window
 * length for the multi-rank window), never more steps than visits */
EXPORT void orc_apply_accum(double* qa, double* qb, double* count, int64_t* accum, const double* alpha_tab, int32_t n_tab, double alpha_min, int per_step,
                            int64_t n_launch) {
  for (int cell = 0; cell < DQL_N_CELLS; ++cell) {
    for (int t = 0; t < 2; ++t) {
      int64_t* acc = accum + t * 2 * DQL_N_CELLS;
      double* q = t ? qb : qa;
      const int64_t m = acc[DQL_N_CELLS + cell];
      if (m <= 0) continue;
      const double tbar = ((double)acc[cell] * (1.0 / (double)(1ll << DQL_TARGET_FRAC_BITS))) / (double)m;
      const int64_t c0 = (int64_t)count[cell];
      double shrink = 1.0; int64_t j = 0;
      const int64_t m_eff = per_step ? (m < n_launch ? m : n_launch) : m;
      for (; j < m_eff && c0 + j < n_tab; ++j) shrink *= (1.0 - alpha_tab[c0 + j]);
      int64_t rem = m_eff - j;
      if (rem > 0) { double base = 1.0 - alpha_min, pw = 1.0; while (rem) { if (rem & 1) pw *= base; base *= base; rem >>= 1; } shrink *= pw; }
      q[cell] = tbar + (q[cell] - tbar) * shrink;
      count[cell] += (double)m;
      acc[cell] = 0; acc[DQL_N_CELLS + cell] = 0;
    }
  }
}
/* physics ticks of agent period j: floor((j+1) T/dt) - floor(j T/dt), T = 1/f_ag */
EXPORT int64_t orc_ticks_before(int64_t j, double f_ag, double dt) { return (int64_t)floor((double)j * (1.0 / (f_ag * dt))); }
#endif
