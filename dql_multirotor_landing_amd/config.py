"""`DqlConfig`: Python mirror of `dql_config` (include/dql.h), field for field, with the reference's defaults.

Every default cites where the reference defines it (paths relative to /root/reference;
pkg = src/dql_multirotor_landing/src/dql_multirotor_landing).  SURVEY.md appendix A lists them all.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List

import numpy as np

MAX_LEVELS = 5
N_ACTIONS = 3
N_ANGLES = 7
STATES_PER_LEVEL = 3 * 3 * 3 * N_ANGLES  # 189
CELLS_PER_LEVEL = STATES_PER_LEVEL * N_ACTIONS  # 567
N_STATES = MAX_LEVELS * STATES_PER_LEVEL  # 945
N_CELLS = MAX_LEVELS * CELLS_PER_LEVEL  # 2835
ACC_LEN = 4 * N_CELLS  # accumulators / exchange window: table a's {target sums, visits}, then table b's (include/dql.h)
TABLE_SHAPE = (MAX_LEVELS, 3, 3, 3, N_ANGLES, N_ACTIONS)
TARGET_FRAC_BITS = 26

F32, F64 = 0, 1
TRAJ_RPM, TRAJ_EIGHT = 0, 1

# quirk switches, same bits as DQL_Q_* in include/dql.h
Q_FAIL_TERM_EVERY_STEP = 1 << 0
Q_STICKY_CHECK = 1 << 1
Q_SHAPING_SURVIVES_RESET = 1 << 2
Q_FROZEN_ACC_REFERENCE = 1 << 3
Q_BOOTSTRAP_ON_POS_CHANGE = 1 << 4
Q_UPDATE_TABLE_A_ONLY = 1 << 5
Q_GOAL_COUNT_KEPT = 1 << 6
Q_REFERENCE = 0x7F
# mode="paper": the reference's code with its reward / observation / update quirks repaired, and its success criterion kept: f_ag
# goal-bin steps at the working level in total (pkg/mdp.py:402-425) — what its trainer promotes on (B17)
Q_PAPER = Q_GOAL_COUNT_KEPT
Q_NONE = 0  # additionally: the success counter restarts whenever the goal bins are left

# CheckResult codes in declaration order of pkg/mdp.py:68-77
CHECK_NAMES = (
    "TERMINAL_CONTACT", "TERMINAL_SUCCESS", "TERMINAL_FLYZONE_X", "TERMINAL_FLYZONE_Y", "TERMINAL_FLYZONE_Z",
    "TERMINAL_MINIMUM_ALTITUDE", "TERMINAL_TIMEOUT", "NON_TERMINAL_SUCCESS", "NON_TERMINAL",
)
N_CHECK_CODES = len(CHECK_NAMES)


class DqlConfigC(C.Structure):
    """ctypes layout of `struct dql_config`."""

    _fields_ = [
        ("working_curriculum_step", C.c_int32), ("two_axis", C.c_int32), ("quirks", C.c_uint32), ("dtype", C.c_int32),
        ("f_ag", C.c_double), ("t_max", C.c_double), ("p_max", C.c_double), ("v_max", C.c_double), ("a_max", C.c_double),
        ("theta_max", C.c_double), ("delta_theta", C.c_double), ("beta", C.c_double), ("sigma_a", C.c_double),
        ("minimum_altitude", C.c_double),
        ("w_p", C.c_double), ("w_v", C.c_double), ("w_theta", C.c_double), ("w_dur", C.c_double), ("w_fail", C.c_double),
        ("w_succ", C.c_double),
        ("lim_p", C.c_double * MAX_LEVELS), ("lim_v", C.c_double * MAX_LEVELS), ("lim_a", C.c_double * MAX_LEVELS),
        ("vz_setpoint", C.c_double), ("yaw_setpoint", C.c_double),
        ("gamma", C.c_double), ("alpha_min", C.c_double), ("alpha_omega", C.c_double),
        ("dt", C.c_double), ("manager_div", C.c_int32), ("trajectory", C.c_int32),
        ("gravity", C.c_double), ("mass", C.c_double), ("inertia", C.c_double * 3),
        ("arm_length", C.c_double), ("rotor_z", C.c_double), ("k_f", C.c_double), ("k_m", C.c_double),
        ("rotor_alpha_up", C.c_double), ("rotor_alpha_down", C.c_double), ("rotor_max", C.c_double),
        ("c_drag", C.c_double), ("c_roll", C.c_double),
        ("k_R", C.c_double * 3), ("k_W", C.c_double * 3),
        ("pid_vz", C.c_double * 6), ("pid_yaw", C.c_double * 6), ("bw_c", C.c_double),
        ("mp_r_x", C.c_double), ("mp_t_x", C.c_double), ("mp_dt", C.c_double),
        ("mp_top_z", C.c_double), ("mp_half_x", C.c_double), ("mp_half_y", C.c_double), ("drone_bottom", C.c_double),
        ("z_init", C.c_double), ("init_sigma", C.c_double), ("init_uniform", C.c_int32), ("per_env_platform", C.c_int32),
        ("goal_logic", C.c_int32), ("fold_per_step", C.c_int32),
        ("mp_r_lo", C.c_double), ("mp_r_hi", C.c_double), ("mp_t_lo", C.c_double), ("mp_t_hi", C.c_double),
        ("noise_pos_sd", C.c_double), ("noise_vel_sd", C.c_double), ("kalman_q", C.c_double),
    ]


def _composite_inertia():
    """Diagonal inertia of base + 4 rotor links about the base origin.

    hummingbird.xacro:29-52 (base 0.68 kg, diag(0.007, 0.007, 0.012); rotor 0.009 kg at arm 0.17 m, 0.01 m above the
    base, rotor box 0.1 x 0.015 x 0.003 m with the mass*slowdown(10) trick of multirotor_base.xacro:24-28).  The rotor
    links spin, so their in-plane box inertia is averaged over a turn.
    """
    m_r, l, h = 0.009, 0.17, 0.01
    mb = m_r * 10.0
    ixx_r = 0.0833333 * mb * (0.015**2 + 0.003**2)
    iyy_r = 0.0833333 * mb * (0.1**2 + 0.003**2)
    izz_r = 0.0833333 * mb * (0.1**2 + 0.015**2)
    inplane = 0.5 * (ixx_r + iyy_r)
    ixx = 0.007 + 2 * m_r * (l * l + h * h) + 2 * m_r * h * h + 4 * inplane
    izz = 0.012 + 4 * m_r * l * l + 4 * izz_r
    return [ixx, ixx, izz]


@dataclass
class DqlConfig:
    # ---- MDP (pkg/mdp.py:87-147; TrainingMdp defaults :214-255) ----
    working_curriculum_step: int = 0
    two_axis: int = 0
    quirks: int = Q_REFERENCE
    dtype: int = F32
    f_ag: float = 22.92  # pkg/trainer.py:42
    t_max: float = 20.0  # pkg/trainer.py:40
    p_max: float = 4.5  # pkg/trainer.py:43
    v_max: float = 3.39411  # pkg/mdp.py:101
    a_max: float = 1.28  # pkg/mdp.py:102
    theta_max: float = float(np.deg2rad(21.37723))  # pkg/mdp.py:103
    delta_theta: float = float(np.deg2rad(7.12574))  # pkg/mdp.py:104
    beta: float = 1 / 3  # pkg/mdp.py:105
    sigma_a: float = 0.416  # pkg/mdp.py:106
    minimum_altitude: float = 0.2  # pkg/mdp.py:234
    w_p: float = -100.0
    w_v: float = -10.0
    w_theta: float = -1.55
    w_dur: float = -6.0
    w_fail: float = -2.6
    w_succ: float = 2.6  # pkg/mdp.py:94-99
    lim_p: List[float] = field(default_factory=lambda: [1.0, 0.64, 0.4096, 0.262144, 0.16777216])  # pkg/mdp.py:45-47
    lim_v: List[float] = field(default_factory=lambda: [1.0, 0.8, 0.64, 0.512, 0.4096])  # :48-50
    lim_a: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0, 1.0, 1.0])  # :51-53
    vz_setpoint: float = -0.1  # pkg/mdp.py:212 (training); -0.4 for the simulation env (:580)
    yaw_setpoint: float = 0.0
    # ---- agent / trainer (pkg/trainer.py:31-33) ----
    gamma: float = 0.99
    alpha_min: float = 0.02949
    alpha_omega: float = 0.51
    # ---- simulator ----
    dt: float = 0.002  # worlds/basic.world:64-70
    manager_div: int = 5  # 100 Hz manager over 500 Hz physics (launch/environment.launch:55)
    trajectory: int = TRAJ_RPM  # launch/environment.launch:60
    gravity: float = 9.8  # worlds/basic.world:36
    mass: float = 0.68 + 4 * 0.009 + 1e-5  # hummingbird.xacro:29,32; mav_generic_odometry_sensor.gazebo:38
    inertia: List[float] = field(default_factory=_composite_inertia)
    arm_length: float = 0.17
    rotor_z: float = 0.01  # hummingbird.xacro:33-34
    k_f: float = 8.54858e-06
    k_m: float = 0.016  # hummingbird.xacro:36-37
    tau_up: float = 0.0125
    tau_down: float = 0.025  # hummingbird.xacro:38-39 (not in the C struct: folded into rotor_alpha_*)
    rotor_max: float = 838.0  # hummingbird.xacro:40
    c_drag: float = 8.06428e-05
    c_roll: float = 1e-06  # hummingbird.xacro:41-42
    k_R: List[float] = field(default_factory=lambda: [0.7, 0.7, 0.035])  # pkg/attitude_controller.py:86
    k_W: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.025])  # :87
    # Kp Ki Kd lower upper windup
    pid_vz: List[float] = field(default_factory=lambda: [5.0, 10.0, 0.0, 0.0, 10.0, 10.0])  # launch/drone.launch:35-40
    pid_yaw: List[float] = field(default_factory=lambda: [8.0, 1.0, 0.0, -3.141592, 3.141592, 5.0])  # :49-54
    bw_c: float = 1.0  # pkg/filters.py:93
    mp_r_x: float = 2.0
    mp_t_x: float = 1.6  # launch/environment.launch:62-65
    mp_dt: float = 0.01  # 1 / frequency (:63)
    mp_top_z: float = 0.455  # urdf/moving_platform.urdf:16,38,51,58 (box top 0.445 + 0.01 bumper)
    mp_half_x: float = 0.55
    mp_half_y: float = 0.55  # 0.5 platform + 0.05 drone base half width (hummingbird.xacro:30)
    drone_bottom: float = 0.06  # half body_height (hummingbird.xacro:31)
    z_init: float = 4.0  # pkg/trainer.py:41
    init_sigma: float = 4.5 / 3  # pkg/landing_simulation_env.py:189
    init_uniform: int = 0  # 0: TrainingLandingEnv.reset (normal at level 0, else uniform); 1: always uniform; 2: SimulationLandingEnv.reset placement
    per_env_platform: int = 0
    goal_logic: int = 1  # 0 = SimulationMdp.check: no goal / success branch (pkg/mdp.py:784-845)
    fold_per_step: int = 0  # 1: one alpha step per launch towards the launch's mean target
    mp_r_lo: float = 1.0
    mp_r_hi: float = 3.0
    mp_t_lo: float = 0.8
    mp_t_hi: float = 1.6
    noise_pos_sd: float = 0.0
    noise_vel_sd: float = 0.0  # launch/environment.launch:56-57
    kalman_q: float = 1e-4  # scripts/manager_node.py:96-98

    # ------------------------------------------------------------------
    def to_c(self) -> DqlConfigC:
        c = DqlConfigC()
        for name, ctype in DqlConfigC._fields_:
            if name == "rotor_alpha_up":
                c.rotor_alpha_up = math.exp(-self.dt / self.tau_up)  # common.h:160
            elif name == "rotor_alpha_down":
                c.rotor_alpha_down = math.exp(-self.dt / self.tau_down)  # common.h:167
            else:
                v = getattr(self, name)
                if isinstance(v, (list, tuple, np.ndarray)):
                    arr = getattr(c, name)
                    if len(v) != len(arr):
                        raise ValueError(f"{name}: expected {len(arr)} values, got {len(v)}")
                    for i, x in enumerate(v):
                        arr[i] = float(x)
                else:
                    setattr(c, name, v)
        return c

    def ticks_before(self, j: int) -> int:
        """Physics ticks elapsed before agent period j: floor(j * T/dt), T = 1/f_ag."""
        return int(math.floor(j * (1.0 / (self.f_ag * self.dt))))

    def alpha_table(self, n: int = 1536) -> np.ndarray:
        """alpha(count) exactly as Trainer.alpha computes it (pkg/trainer.py:88-110): count 0 -> alpha_min,
        else max(float_power(1/count, omega), alpha_min).  The table must reach the alpha_min plateau (1 536 entries do for the
        reference's alpha_min; smaller floors get as many as they need)."""
        if 0 < self.alpha_min < 1 and self.alpha_omega > 0:
            n = max(n, min(1 << 22, int(self.alpha_min ** (-1.0 / self.alpha_omega)) + 8))
        tab = np.empty(n, dtype=np.float64)
        tab[0] = self.alpha_min
        for cnt in range(1, n):
            tab[cnt] = float(np.max([np.float_power(1 / np.float64(cnt), self.alpha_omega), self.alpha_min]))
        if tab[-1] != self.alpha_min:
            raise ValueError("alpha table too short: plateau alpha_min not reached")
        return tab


def training_config(level: int = 0, **kw) -> DqlConfig:
    """TrainingLandingEnv as the Trainer builds it (pkg/trainer.py:176-183)."""
    return DqlConfig(working_curriculum_step=level, **kw)


AS_LAUNCHED = dict(mp_t_x=1.0, noise_pos_sd=0.25, noise_vel_sd=0.1)  # as_launched_config's fields, for Trainer(env_kw=AS_LAUNCHED)


def as_launched_config(level: int = 0, **kw) -> DqlConfig:
    """The parameters the reference's manager node actually RAN with under `roslaunch` — not the ones its launch file spells out.

    `launch/environment.launch:54-72` starts the node as `<node name="manager_node" …>` and sets its private parameters (noise 0, t_x 1.6)
    under `/hummingbird/manager_node/…`; roslaunch's `__name:=manager_node` overrides `rospy.init_node("central_logic_node")`, but the code
    reads `ns + "central_logic_node/…"` (`scripts/manager_node.py:73-91`, `pkg/moving_platform.py:49-69`), finds nothing there and takes its
    in-code defaults: observation noise 0.25 m / 0.1 m/s (Kalman R = 0.1^2) and platform speed t_x = 1 m/s (r_x = 2 m: omega = 0.5 rad/s).
    The PID nodes read `~`-private names and are not affected (`pkg/pid.py:33-48`).  The reference's own Gazebo flight records
    decide between the two readings (tests/test_g14_gazebo.py, golden G14): 801 episodes at eps = 1 reach the goal state in 38.0 %; this
    simulator gives 37.0 % with these values and 27.3 % with the launch file's, far outside the 99 % interval."""
    base = dict(working_curriculum_step=level, **AS_LAUNCHED)
    base.update(kw)
    return DqlConfig(**base)


def simulation_config(**kw) -> DqlConfig:
    """SimulationLandingEnv defaults (pkg/landing_simulation_env.py:285-306; pkg/mdp.py:580): level 4, v_z -0.4, uniform start offset placed
    as its reset() does (clip(platform - offset, +-p_max), :331-343; pinned by tests/golden G13), no goal / success branch in check()."""
    base = dict(working_curriculum_step=4, vz_setpoint=-0.4, init_uniform=2, z_init=4.0, goal_logic=0)
    base.update(kw)
    return DqlConfig(**base)
