"""The tick's control-side functions against the golden vectors the reference's OWN Python produced (G8 filters / PID, G9 attitude law,
G11 platform, G12 manager tick), for any backend that exposes the replay operators — the CPU oracle (tests/test_oracle_golden.py,
tests/test_f32_fixtures.py) and the HIP library through its C ABI (tests/test_gpu_operators.py) — and in BOTH dtypes:

* float64 spells the reference's expressions out operation by operation: bit-exact (or the fixture's own stand-in tolerance).
* float32 — the arithmetic every throughput figure runs on — takes the same formulas in shorter forms (transposed Butterworth, fused
  multiply-adds, x-axis attitude closed form, v_rsq + residual-correction root behind one med3 clamp, platform sine / cosine carried by rotation,
  Kalman fixed-point shortcut, lazy noise).  Each check below states the bound it asserts and where the bound comes from; eps = 2^-24
  (float32 half ulp relative).  The bounds are a priori (rounding analysis), the measured maxima are quoted next to them.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from dql_multirotor_landing_amd.config import DqlConfig, F32, F64, Q_REFERENCE, TRAJ_EIGHT

GOLDEN = Path(__file__).resolve().parent / "golden"
EPS = 2.0 ** -24


class OracleBackend:
    """oracle/oracle.py's replay operators behind the interface the checks use"""
    name = "oracle"

    def __init__(self):
        from oracle import oracle as orc
        self.o = orc

    def butterworth_run(self, cfg, x):
        return self.o.butterworth_run(x, c=cfg.bw_c, dtype=cfg.dtype)

    def kalman_run(self, cfg, vel, flags):
        return self.o.kalman_run(vel, flags, cfg.kalman_q, cfg.noise_vel_sd, dtype=cfg.dtype)

    def pid_run(self, cfg, params, state):
        return self.o.pid_run(params, state, bw_c=cfg.bw_c, dtype=cfg.dtype)

    def attitude_run(self, cfg, q, w, cmd, xonly=0):
        return self.o.attitude_rotors(cfg, q, w, cmd, xonly=xonly)

    def platform_run(self, cfg, n, carry=0):
        return self.o.platform_run(cfg, n, dtype=cfg.dtype, carry=carry)

    def manager_run(self, cfg, series, contact, seed=0):
        return self.o.manager_run(cfg, series, contact, seed=seed)


class HipBackend:
    """the product's stateless operators (dql_*_run through ctypes: dql_multirotor_landing_amd/ops.py)"""
    name = "hip"

    def __init__(self):
        from dql_multirotor_landing_amd import ops
        self.o = ops

    def butterworth_run(self, cfg, x):
        return self.o.butterworth_run(cfg, x)

    def kalman_run(self, cfg, vel, flags):
        return self.o.kalman_run(cfg, vel, flags)

    def pid_run(self, cfg, params, state):
        return self.o.pid_run(cfg, params, state)

    def attitude_run(self, cfg, q, w, cmd, xonly=0):
        return self.o.attitude_run(cfg, q, w, cmd, xonly=xonly)

    def platform_run(self, cfg, n, carry=0):
        return self.o.platform_run(cfg, n, carry=carry)

    def manager_run(self, cfg, series, contact, seed=0):
        return self.o.manager_run(cfg, np.asarray(series)[None], np.asarray(contact)[None], seed=seed)[0]


# ---------------------------------------------------------------------------------------------------------------------------------
# G8: pkg/filters.py:98-109 (Butterworth), :19-80 (Kalman), pkg/pid.py:62-104 (PID.output)
# ---------------------------------------------------------------------------------------------------------------------------------
def check_g8_butterworth(be, dtype):
    g = np.load(GOLDEN / "g8_filters.npz")
    y = be.butterworth_run(DqlConfig(dtype=dtype), g["bw_in"])
    if dtype == F64:
        np.testing.assert_array_equal(y, g["bw_out"])
        return 0.0
    # float32, transposed direct form: per output 4 roundings of values <= 2 |x|max; the filter is a unit-DC-gain low-pass (poles inside
    # 0.42) so the rounding noise is not amplified: |err| <= 16 eps |x|max = 2.5e-6 for |x| <= 2.64.  Measured: 2.0e-7.
    bound = 16 * EPS * np.abs(g["bw_in"]).max()
    err = np.abs(y - g["bw_out"]).max()
    assert err <= bound, (err, bound)
    return err


def check_g8_kalman(be, dtype):
    g = np.load(GOLDEN / "g8_filters.npz")
    worst = 0.0
    for tag, sd in (("r0", 0.0), ("r01", 0.1)):
        vel = g[f"kf_vel_{tag}"]
        flags = np.array([(i % 17 == 0) for i in range(len(vel))], dtype=np.uint8)
        acc = be.kalman_run(DqlConfig(dtype=dtype, noise_vel_sd=sd, kalman_q=1e-4), vel, flags)
        ref = g[f"kf_acc_{tag}"]
        if dtype == F64:
            np.testing.assert_array_equal(acc, ref)
            continue
        # float32: z = dv / dt.  dv: two roundings of |v| <= 0.55 -> 2 eps 0.55 / 0.01 = 6.6e-6 absolute; dt = difference of two float32
        # time stamps <= 1.2 s -> 2 eps 1.2 / 0.01 = 1.4e-5 RELATIVE (the reference subtracts time stamps too, in float64); the filter
        # (gain <= 1) does not amplify.  Measured: 1.2e-4 at |acc| = 15.
        bound = 1e-5 + 2e-5 * np.abs(ref)
        assert (np.abs(acc - ref) <= bound).all(), (tag, np.abs(acc - ref).max())
        worst = max(worst, np.abs(acc - ref).max())
    return worst


def check_g8_pid(be, dtype):
    g = np.load(GOLDEN / "g8_filters.npz")
    worst = 0.0
    for tag in ("vz", "yaw"):  # the third fixture has Kd != 0: the reference launches both controllers with Kd = 0 and the kernel has no D term
        params, state = g[f"pid_{tag}_params"], g[f"pid_{tag}_state"]
        eff, integ = be.pid_run(DqlConfig(dtype=dtype), params, state)
        if dtype == F64:
            np.testing.assert_array_equal(integ, g[f"pid_{tag}_integral"])
            np.testing.assert_array_equal(eff, g[f"pid_{tag}_effort"])
            continue
        # float32: the integral is a running sum of e dt: one rounding (eps |I|, |I| <= windup) per tick, 400 ticks, and dt itself is a
        # difference of float32 time stamps (relative 2 eps 0.8 / 0.002 = 5e-5 of each increment |e| dt <= 0.01): <= 400 (eps 1 + 5e-7) = 2.3e-4
        # worst case; measured 2.0e-7.  Effort = Kp fe + Ki I: Kp (Butterworth bound) + Ki (integral bound) + 2 roundings.
        kp, ki, wind = params[0], params[1], params[5]
        scale_e = np.abs(params[6] - state).max()
        b_int = 400 * (EPS * min(wind, 400 * 0.002 * scale_e) + 5e-5 * 0.002 * scale_e)
        b_eff = kp * 16 * EPS * scale_e + ki * b_int + 4 * EPS * max(abs(params[3]), abs(params[4]))
        e_int = np.abs(integ - g[f"pid_{tag}_integral"]).max(); e_eff = np.abs(eff - g[f"pid_{tag}_effort"]).max()
        assert e_int <= b_int and e_eff <= b_eff, (tag, e_int, b_int, e_eff, b_eff)
        ref_eff = g[f"pid_{tag}_effort"]
        for lim in (params[3], params[4]):  # a saturated output is the limit itself — its float32 cast — bit for bit (med3 clamp)
            sat = ref_eff == lim
            assert (eff[sat] == float(np.float32(lim))).mean() > 0.98 if sat.any() else True
        worst = max(worst, e_int, e_eff)
    return worst


# ---------------------------------------------------------------------------------------------------------------------------------
# G9: pkg/attitude_controller.py:107-156 -> commanded rotor speeds
# ---------------------------------------------------------------------------------------------------------------------------------
def _g9():
    g = np.load(GOLDEN / "g9_attitude.npz")
    x, y, z, w = g["quat_xyzw"].T
    tilt = np.degrees(np.arccos(np.sqrt(np.clip((1 - 2 * (y * y + z * z)) ** 2 + (2 * (x * y + w * z)) ** 2, 0, 1))))
    return g, tilt


def _alloc_scale(cfg, thrust, moment):
    """size of the terms the inverse allocation adds up to a rotor's w^2 = ia T -+ ib M_xy +- ic M_z (attitude_controller.py:94-105, 148-156)"""
    ia, ib, ic = 1 / (4 * cfg.k_f), 1 / (2 * cfg.arm_length * cfg.k_f), 1 / (4 * cfg.k_f * cfg.k_m)
    return ia * np.abs(thrust) + ib * np.maximum(np.abs(moment[:, 0]), np.abs(moment[:, 1])) + ic * np.abs(moment[:, 2])


def check_g9_rotor_speeds(be, dtype):
    """The reference's rotor speeds sqrt(max(w^2, 0)) for 200 (attitude, body rate, command) samples, incl. the clamp at 0 (112 entries) —
    for float32 through the whole short form: quat_to_R by doubled components, Newton yaw frame, yaw-free attitude error, gains with the
    halving folded in, fused inverse allocation, med3 clamp, v_rsq + residual-correction square root."""
    g, tilt = _g9()
    cfg = DqlConfig(dtype=dtype)
    rot = be.attitude_run(cfg, g["quat_xyzw"], g["omega"], g["cmd"])
    ref = g["rotor"]
    if dtype == F64:
        np.testing.assert_allclose(rot, ref, rtol=1e-10, atol=1e-7)
        return 0.0
    # float32: compared as w^2 (what the law computes; the root of a small w^2 magnifies any error by 1 / (2 w)).  Every w^2 is a sum of
    # three terms of size S (_alloc_scale) carrying ~20 roundings each through R, E, e_R, M: <= 1e-6 S up to 55 deg of tilt (measured
    # 4.0e-7 S); beyond, the yaw frame's 1 / sqrt by three Newton steps from a second-order start is the larger term: 5e-5 S (measured
    # 1.5e-5 S at the fixture's 60 deg samples; the reference's envs fly within +-22 deg, mdp.py theta_max)
    S = _alloc_scale(cfg, g["cmd"][:, 3], g["moment"])
    err = np.abs(rot ** 2 - ref ** 2).max(axis=1) / S
    assert (tilt <= 55).sum() >= 190 and (tilt > 55).sum() >= 3
    assert err[tilt <= 55].max() <= 1e-6, err[tilt <= 55].max()
    assert err[tilt > 55].max() <= 5e-5, err[tilt > 55].max()
    # the clamp at zero: the reference commands exactly 0, the float32 form 1e-15 rad/s (sqrt of the med3 floor 1e-30: sqrt_pos's domain)
    z = ref == 0.0
    assert z.sum() > 100 and (rot[z] <= 1.0000001e-15).mean() > 0.97  # (a w^2 within rounding of 0 may land on the other side)
    assert rot.max() <= cfg.rotor_max
    return err.max()


def check_g9_xonly_form(be):
    """The x-axis kernels' closed form (roll command exactly 0) against the generic float32 form and against the reference-pinned float64
    law on the fixture's attitudes with the roll command zeroed: same law, 10 instructions shorter."""
    g, tilt = _g9()
    cmd = g["cmd"].copy(); cmd[:, 0] = 0.0
    c32, c64 = DqlConfig(dtype=F32), DqlConfig(dtype=F64)
    rx = be.attitude_run(c32, g["quat_xyzw"], g["omega"], cmd, xonly=1)
    rg = be.attitude_run(c32, g["quat_xyzw"], g["omega"], cmd, xonly=0)
    r64 = be.attitude_run(c64, g["quat_xyzw"], g["omega"], cmd)
    # scale from the float64 law's own moments: recover them from w^2 by the allocation matrix A (fixture "A") where no rotor is clamped
    w2 = r64 ** 2
    tm = w2 @ g["A"].T  # rows: roll, pitch, yaw moment, thrust (attitude_controller.py:94-105)
    S = _alloc_scale(c64, cmd[:, 3], np.abs(tm[:, :3])) + 1.0
    ok = tilt <= 55
    assert (np.abs(rx ** 2 - w2).max(axis=1) / S)[ok].max() <= 2e-6
    assert (np.abs(rx ** 2 - rg ** 2).max(axis=1) / S)[ok].max() <= 2e-6  # B01 = B10 = 0 and B11 = 1 exactly: only signed zeros and fusion differ


# ---------------------------------------------------------------------------------------------------------------------------------
# G11: pkg/moving_platform.py:87-127
# ---------------------------------------------------------------------------------------------------------------------------------
G11_CASES = (("rpm_launch", {}), ("rpm_default", dict(mp_t_x=1.0)), ("eight", dict(trajectory=TRAJ_EIGHT)))


def check_g11_platform(be, dtype, carry=0):
    g = np.load(GOLDEN / "g11_platform.npz")
    worst = 0.0
    for name, kw in G11_CASES:
        cfg = DqlConfig(dtype=dtype, **kw)
        out = be.platform_run(cfg, 3000, carry=carry)
        ref = g[name][:, 1:]
        if dtype == F64:
            np.testing.assert_allclose(out, ref, rtol=0, atol=2e-11)
            continue
        # float32: the phase is the state, advanced by fma(omega, dt, phase) and wrapped at 2 pi: one rounding of <= 2^-22 (half an ulp in
        # [4, 8)) per tick, and for a CONSTANT increment the roundings inside one binade share a sign, so the bound is linear in the tick
        # index: |d phase_i| <= (i + 1) 2^-22 + 2 eps 2 pi (wraps).  x = r sin: |dx| <= r |d phase| + 4 eps r (polynomial sine, product);
        # u = r omega cos likewise.  The carried variant (sine / cosine re-seeded every `carry` ticks and rotated in between) adds <= 4 eps
        # per rotation, at most `carry` - 1 of them in a row.  Measured after 30 s: 3.4e-4 m of a bound of 1.4e-3 (eight: 2.1e-3).
        r = 3.0 if kw.get("trajectory") == TRAJ_EIGHT else cfg.mp_r_x
        om = 0.8 / 3.0 if kw.get("trajectory") == TRAJ_EIGHT else cfg.mp_t_x / cfg.mp_r_x
        i = np.arange(3000) + 1.0
        dph = i * 2.0 ** -22 + 4 * EPS * np.pi + (4 * EPS * max(carry - 1, 0))
        bx = r * dph + 8 * EPS * r
        bound = np.stack([bx, 2 * bx, om * bx, 2 * om * bx], axis=1)  # eight: y = r sin cos, v = r omega cos 2 phase: twice the sensitivity
        err = np.abs(out - ref)
        assert (err <= bound).all(), (name, carry, (err / bound).max())
        if kw.get("trajectory") != TRAJ_EIGHT:
            assert (out[:, 1] == 0).all() and (out[:, 3] == 0).all()
        worst = max(worst, err.max())
    return worst


# ---------------------------------------------------------------------------------------------------------------------------------
# G12: scripts/manager_node.py:192-214, 292-310 + pkg/observation_utils.py:77-158
# ---------------------------------------------------------------------------------------------------------------------------------
def g12_cfg(noise_sd, dtype, quirks=Q_REFERENCE):
    return DqlConfig(dtype=dtype, two_axis=1, noise_pos_sd=float(noise_sd[0]), noise_vel_sd=float(noise_sd[1]), quirks=quirks)


def check_g12_manager_f32(be):
    """The float32 manager tick against ManagerNode.publish_obs over the fixture's three 300-tick series (noise 0, yaw != 0, noise > 0 with
    the Kalman gain at R = sd_v^2).  Columns: obs p_x p_y v_x v_y a_x a_y, v_z plant state, yaw plant state, platform set-point x y u v."""
    z = np.load(GOLDEN / "g12_manager.npz")
    worst = {}
    for tag in ("a_noise0", "c_yaw", "b_noise"):
        got = be.manager_run(g12_cfg(z[f"{tag}_noise_sd"], F32), z[f"{tag}_in"], z[f"{tag}_contact"], seed=5)
        ref = z[f"{tag}_out"]
        inp = z[f"{tag}_in"]
        i = np.arange(len(ref), dtype=np.float64)
        if tag != "b_noise":  # with noise the published p / v carry this build's Philox draws instead of numpy's: compared in distribution (test_g12_noise_*)
            # p / v: differences and a 2x2 rotation of float32 casts of |p| <= 4.5 m, |v| <= 2 m/s: 6 roundings -> 6 eps 4.5 = 1.6e-6
            assert np.abs(got[:, :4] - ref[:, :4]).max() <= 1.6e-6, np.abs(got[:, :4] - ref[:, :4]).max()
        # acceleration under B19 (frozen reference sample): (v_i - v_0) / (0.01 i); v_i and v_0 are each a 2x2 rotation of a difference of
        # two float32 casts, 4 roundings of |v| <= 2 m/s -> 8 eps each, 16 eps / (0.01 i) together, and the float32 product i * 0.01 as the
        # divisor: 2 eps relative; the Kalman gain (<= 1) does not amplify.  Tick 0 publishes 0 exactly.
        assert (got[0, 4:6] == 0).all()
        b_in = 16 * EPS / (0.01 * np.maximum(i, 1.0)) + 4 * EPS * np.abs(ref[:, 4:6]).max(axis=1) + 1e-6
        # ... per tick INPUT of the filter; with R > 0 the estimate remembers earlier inputs: e_i <= (1 - K_i) e_(i-1) + K_i b_i with the
        # filter's own (data-independent) gains K_i (pkg/filters.py:19-36: P += Q; K = P / (P + R); P *= 1 - K)
        P, Q, Rm = 1.0, 1e-4, float(z[f"{tag}_noise_sd"][1]) ** 2
        b_acc = np.zeros(len(ref))
        for k in range(1, len(ref)):
            P += Q; K = 1.0 if Rm == 0.0 else P / (P + Rm); P *= 1 - K
            b_acc[k] = (1 - K) * b_acc[k - 1] + K * b_in[k]
        b_acc += 1e-6
        e_acc = np.abs(got[:, 4:6] - ref[:, 4:6]).max(axis=1)
        assert (e_acc <= b_acc).all(), (tag, (e_acc / b_acc).max())
        # PID plant states: v_z is a cast (eps |v_z|); yaw = atan2 of the yaw-only frame: fdlibm's float kernel, < 1e-6 rad
        assert np.abs(got[:, 6] - ref[:, 6]).max() <= 2 * EPS * np.abs(inp[:, 5]).max() + 1e-9
        assert np.abs(got[:, 7] - ref[:, 7]).max() <= 1e-6
        # the platform set-point published by the tick: G11's linear bound over 300 ticks
        bsp = 2.0 * ((i + 1) * 2.0 ** -22 + 4 * EPS * np.pi) + 16 * EPS
        assert (np.abs(got[:, 8] - ref[:, 8]) <= bsp).all() and (np.abs(got[:, 10] - ref[:, 10]) <= bsp).all()
        worst[tag] = (float(e_acc.max()), float(np.abs(got[:, 8] - ref[:, 8]).max()))
    return worst
