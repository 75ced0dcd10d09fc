"""Stateless batch operators of the C ABI (computed on the device): the arithmetic behind the drop-in
`TrainingMdp` / `DoubleQLearningAgent` methods."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .config import DqlConfig, Q_REFERENCE


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def discretise(cfg: DqlConfig, rel_p, rel_v, rel_a, angle, device: int = 0) -> np.ndarray:
    """TrainingMdp.discrete_state (pkg/mdp.py:257-333) for n observations -> packed indices (-1 where the reference raises)."""
    p, v, a, t = map(_f64, (rel_p, rel_v, rel_a, angle))
    if not (p.shape == v.shape == a.shape == t.shape and p.ndim == 1):
        raise ValueError("inputs must be 1-D arrays of equal length")
    out = np.zeros(len(p), dtype=np.int32)
    c = cfg.to_c()
    _lib.check(_lib.load().dql_discretise(C.byref(c), device, _p(p), _p(v), _p(a), _p(t), len(p), _p(out)))
    return out


MDP_ACTION, MDP_DISCRETISE, MDP_CHECK, MDP_REWARD, MDP_SIMULATION, MDP_ALL = 1, 2, 4, 8, 16, 15


def mdp_transition(cfg: DqlConfig, action, obs, mdp_state, prev_idx, idx=None, stages: int = MDP_ALL, device: int = 0):
    """Selected TrainingMdp / SimulationMdp methods (see DQL_MDP_* in include/dql.h) for n independent MDPs.
    Returns (mdp_state, idx, reward, done); arrays not touched by the selected stages come back unchanged / zero."""
    n = len(action)
    action = np.ascontiguousarray(action, dtype=np.uint8)
    obs = _f64(obs); ms = _f64(mdp_state).copy(); prev_idx = np.ascontiguousarray(prev_idx, dtype=np.int32)
    if obs.shape != (7, n) or ms.shape != (8, n) or prev_idx.shape != (n,):
        raise ValueError("obs must be [7][n], mdp_state [8][n], prev_idx [n]")
    idx = np.full(n, -1, dtype=np.int32) if idx is None else np.ascontiguousarray(idx, dtype=np.int32).copy()
    rew = np.zeros(n); done = np.zeros(n, dtype=np.uint8)
    c = cfg.to_c()
    _lib.check(_lib.load().dql_mdp_transition(C.byref(c), device, n, stages, _p(action), _p(obs), _p(ms), _p(prev_idx), _p(idx), _p(rew), _p(done)))
    return ms, idx, rew, done


def manager_run(cfg: DqlConfig, series, contact, seed: int = 0, device: int = 0) -> np.ndarray:
    """The 100 Hz manager tick (ManagerNode.publish_obs + ObservationUtils) over scripted series: `series` [n_series][n_ticks][14]
    (drone p, v, quaternion wxyz, platform x y u v), `contact` [n_series][n_ticks] -> [n_series][n_ticks][12] (include/dql.h)."""
    a = _f64(series); c = np.ascontiguousarray(contact, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 14 or c.shape != a.shape[:2]:
        raise ValueError("series must be [n_series][n_ticks][14], contact [n_series][n_ticks]")
    out = np.zeros(a.shape[:2] + (12,))
    cc = cfg.to_c()
    _lib.check(_lib.load().dql_manager_run(C.byref(cc), device, a.shape[0], a.shape[1], _p(a), _p(c), int(seed), _p(out)))
    return out


def plant_run(cfg: DqlConfig, init, rotor_cmd, device: int = 0) -> np.ndarray:
    """The plant of the fused step (rotor forces + rigid body + rotor filter + platform extrapolation / contact latch: what stands in
    for gazebo_motor_model.cpp + ODE) open loop: `init` [n_series][21] (p, v, quaternion wxyz, body rates, rotor speeds, platform
    x y u v), `rotor_cmd` [n_series][n_ticks][4] -> [n_series][n_ticks][20] state after every 500 Hz tick (include/dql.h)."""
    a = _f64(init); b = _f64(rotor_cmd)
    if a.ndim != 2 or a.shape[1] != 21 or b.ndim != 3 or b.shape[2] != 4 or b.shape[0] != a.shape[0]:
        raise ValueError("init must be [n_series][21], rotor_cmd [n_series][n_ticks][4]")
    out = np.zeros(b.shape[:2] + (20,))
    cc = cfg.to_c()
    _lib.check(_lib.load().dql_plant_run(C.byref(cc), device, b.shape[0], b.shape[1], _p(a), _p(b), _p(out)))
    return out


def butterworth_run(cfg: DqlConfig, x, device: int = 0) -> np.ndarray:
    """ButterworthFilter.update (pkg/filters.py:98-109) over a series from zero histories, in cfg.dtype arithmetic."""
    x = _f64(x)
    if x.ndim != 1:
        raise ValueError("x must be 1-D")
    y = np.zeros_like(x)
    c = cfg.to_c()
    _lib.check(_lib.load().dql_butterworth_run(C.byref(c), device, _p(x), len(x), _p(y)))
    return y


def kalman_run(cfg: DqlConfig, vel, dt_le0, device: int = 0) -> np.ndarray:
    """KalmanFilter3D.filter (pkg/filters.py:53-80) over vel [n][3] sampled at 100 Hz -> acceleration estimates [n - 1][3]
    (Q = cfg.kalman_q, R = cfg.noise_vel_sd ** 2; dt_le0[i] forces the reference's dt <= 0 branch)."""
    vel = _f64(vel); flags = np.ascontiguousarray(dt_le0, dtype=np.uint8)
    if vel.ndim != 2 or vel.shape[1] != 3 or flags.shape != (len(vel),):
        raise ValueError("vel must be [n][3], dt_le0 [n]")
    acc = np.zeros((max(len(vel) - 1, 0), 3))
    c = cfg.to_c()
    _lib.check(_lib.load().dql_kalman_run(C.byref(c), device, _p(vel), _p(flags), len(vel), _p(acc)))
    return acc


def pid_run(cfg: DqlConfig, params, state, device: int = 0):
    """PID.output (pkg/pid.py:62-104) over 500 Hz ticks; params = Kp Ki Kd lower upper windup setpoint (Kd = 0) -> (effort, integral)."""
    params = _f64(params); state = _f64(state)
    if params.shape != (7,) or state.ndim != 1:
        raise ValueError("params must have 7 entries, state must be 1-D")
    eff = np.zeros(len(state)); integ = np.zeros(len(state))
    c = cfg.to_c()
    _lib.check(_lib.load().dql_pid_run(C.byref(c), device, _p(params), _p(state), len(state), _p(eff), _p(integ)))
    return eff, integ


def attitude_run(cfg: DqlConfig, quat_xyzw, omega, cmd, xonly: int = 0, device: int = 0) -> np.ndarray:
    """AttitudeController.compute_rotor_velocities (pkg/attitude_controller.py:107-156): quaternions (x, y, z, w), body rates,
    cmd = roll, pitch, yaw rate, thrust -> commanded rotor speeds [n][4]."""
    q, w, u = map(_f64, (quat_xyzw, omega, cmd))
    n = len(q)
    if q.shape != (n, 4) or w.shape != (n, 3) or u.shape != (n, 4):
        raise ValueError("quat_xyzw must be [n][4], omega [n][3], cmd [n][4]")
    rot = np.zeros((n, 4))
    c = cfg.to_c()
    _lib.check(_lib.load().dql_attitude_run(C.byref(c), device, _p(q), _p(w), _p(u), n, int(xonly), _p(rot)))
    return rot


def platform_run(cfg: DqlConfig, n: int, carry: int = 0, device: int = 0) -> np.ndarray:
    """MovingPlatform.compute_trajectory (pkg/moving_platform.py:87-127) from phase 0 -> [n][4] = x, y, u, v per 100 Hz tick."""
    out = np.zeros((int(n), 4))
    c = cfg.to_c()
    _lib.check(_lib.load().dql_platform_run(C.byref(c), device, int(n), int(carry), _p(out)))
    return out


def selftest_sqrt(lo: float = 1e-30, hi: float = 3.4028234663852886e38, device: int = 0) -> int:
    """number of float32 inputs in [lo, hi] for which the float32 tick's square root is not the correctly rounded one (must be 0)"""
    lo_b = int(np.float32(lo).view(np.uint32)); hi_b = int(np.float32(hi).view(np.uint32))
    n = C.c_int64(-1)
    _lib.check(_lib.load().dql_diag_selftest_sqrt(device, lo_b, hi_b, C.byref(n)))
    return n.value


def place(cfg: DqlConfig, x0, mp, device: int = 0) -> np.ndarray:
    """Drone start coordinate for (random offset, platform coordinate) pairs: the reset placement selected by cfg.init_uniform."""
    x0, mp = _f64(x0), _f64(mp)
    if x0.shape != mp.shape or x0.ndim != 1:
        raise ValueError("x0 and mp must be 1-D arrays of equal length")
    out = np.zeros(len(x0))
    cc = cfg.to_c()
    _lib.check(_lib.load().dql_place(C.byref(cc), device, _p(x0), _p(mp), len(x0), _p(out)))
    return out


def agent_transfer(qa, qb, k: int, ratio: float, device: int = 0):
    """In-place DoubleQLearningAgent.transfer_learning on contiguous float64 tables of 2835 cells."""
    for t in (qa, qb):
        if t.dtype != np.float64 or not t.flags.c_contiguous or t.size != 2835:
            raise ValueError("tables must be contiguous float64 arrays of 2835 cells")
    _lib.check(_lib.load().dql_agent_transfer(device, _p(qa), _p(qb), int(k), float(ratio)))


def agent_predict(qa, qb, idx, device: int = 0) -> np.ndarray:
    qa, qb = _f64(qa).ravel(), _f64(qb).ravel()
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros(len(idx), dtype=np.uint8)
    _lib.check(_lib.load().dql_agent_predict(device, _p(qa), _p(qb), _p(idx), len(idx), _p(out)))
    return out


def agent_update(qa, qb, count, sa, ns, alpha, gamma, reward, quirks: int = Q_REFERENCE, device: int = 0, coin=None, done=None):
    """In-place ordered replay of DoubleQLearningAgent.update; qa/qb/count must be contiguous float64 of 2835 cells.
    `quirks`: Q_REFERENCE reproduces the reference (table a only, bootstrap on a position-bin change); with
    Q_UPDATE_TABLE_A_ONLY cleared it is Double Q-learning and needs `coin` (0 = update table a, 1 = table b) per transition,
    with Q_BOOTSTRAP_ON_POS_CHANGE cleared it needs the `done` flags (include/dql.h)."""
    for t in (qa, qb, count):
        if t.dtype != np.float64 or not t.flags.c_contiguous or t.size != 2835:
            raise ValueError("tables must be contiguous float64 arrays of 2835 cells")
    sa = np.ascontiguousarray(sa, dtype=np.int32); ns = np.ascontiguousarray(ns, dtype=np.int32)
    alpha = _f64(alpha); reward = _f64(reward)
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
    coin, done = u8(coin), u8(done)
    for a in (coin, done):
        if a is not None and a.shape != sa.shape:
            raise ValueError("coin / done must have one entry per transition")
    _lib.check(_lib.load().dql_agent_update(device, _p(qa), _p(qb), _p(count), _p(sa), _p(ns), _p(alpha), float(gamma), _p(reward), len(sa), quirks,
                                            None if coin is None else _p(coin), None if done is None else _p(done)))
