"""Drop-in mirror of the reference's `mdp.py` (pkg/mdp.py): same class names, constructor keywords, method names,
return values and exception types.  The arithmetic of every method runs on the device through the C ABI
(`dql_discretise` / `dql_mdp_transition`, include/dql.h); this module only keeps the handful of per-MDP scalars
between calls, formats the `info` dictionary and raises where the reference raises.

Differences by construction: `Action` / `Observation` are plain Python classes (the reference's are catkin-generated
ROS messages, src/dql_multirotor_landing/msg/*.msg)."""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import ops
from .config import DqlConfig, F64, Q_REFERENCE

State = Tuple[int, int, int, int, int]


class Action:
    """msg/Action.msg: roll, pitch, yaw, v_z."""

    def __init__(self, roll: float = 0.0, pitch: float = 0.0, yaw: float = 0.0, v_z: float = 0.0) -> None:
        self.roll, self.pitch, self.yaw, self.v_z = roll, pitch, yaw, v_z

    def __repr__(self):
        return f"Action(roll={self.roll}, pitch={self.pitch}, yaw={self.yaw}, v_z={self.v_z})"


class Observation:
    """msg/Observation.msg: relative position / velocity / acceleration + contact flag."""

    def __init__(self, rel_p_x=0.0, rel_p_y=0.0, rel_p_z=0.0, rel_v_x=0.0, rel_v_y=0.0, rel_v_z=0.0,
                 rel_a_x=0.0, rel_a_y=0.0, rel_a_z=0.0, contact=False) -> None:
        self.rel_p_x, self.rel_p_y, self.rel_p_z = rel_p_x, rel_p_y, rel_p_z
        self.rel_v_x, self.rel_v_y, self.rel_v_z = rel_v_x, rel_v_y, rel_v_z
        self.rel_a_x, self.rel_a_y, self.rel_a_z = rel_a_x, rel_a_y, rel_a_z
        self.contact = contact


class ContinuousObservation:
    """pkg/mdp.py:11-32."""

    def __init__(self, observation: Optional[Observation] = None, pitch: float = 0.0, roll: float = 0.0,
                 abs_p_z: float = 0.0, contact: bool = False) -> None:
        observation = Observation() if observation is None else observation
        for f in ("rel_p_x", "rel_p_y", "rel_p_z", "rel_v_x", "rel_v_y", "rel_v_z", "rel_a_x", "rel_a_y", "rel_a_z", "contact"):
            setattr(self, f, getattr(observation, f))
        self.pitch, self.roll, self.abs_p_z = pitch, roll, abs_p_z


@dataclass
class RewardShapingValue:
    position: float = 0
    velocity: float = 0
    angle: float = 0


@dataclass
class Limits:
    """pkg/mdp.py:42-65."""
    _working_curriculum_step: int
    _position: List[float] = field(default_factory=lambda: [1.0, 0.64, 0.4096, 0.262144, 0.16777216])
    _velocity: List[float] = field(default_factory=lambda: [1.0, 0.8, 0.64, 0.512, 0.4096])
    _acceleration: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0, 1.0, 1.0])

    @property
    def position(self) -> List[float]:
        return self._position[: self._working_curriculum_step + 1]

    @property
    def velocity(self) -> List[float]:
        return self._velocity[: self._working_curriculum_step + 1]

    @property
    def acceleration(self) -> List[float]:
        return self._acceleration[: self._working_curriculum_step + 1]


class CheckResult(enum.Enum):
    """pkg/mdp.py:68-77 (declaration order = the integer codes of include/dql.h)."""
    TERMINAL_CONTACT = "SUCCESS: Touched platform"
    TERMINAL_SUCCESS = "SUCCESS: Goal state reached"
    TERMINAL_FLYZONE_X = "FAILURE: Drone moved too far from platform in x direction"
    TERMINAL_FLYZONE_Y = "FAILURE: Drone moved too far from platform in y direction"
    TERMINAL_FLYZONE_Z = "FAILURE: Drone moved too far from platform in z direction"
    TERMINAL_MINIMUM_ALTITUDE = "FAILURE: Reached minimum altitude"
    TERMINAL_TIMEOUT = "FAILURE: Maximum episode duration"
    NON_TERMINAL_SUCCESS = enum.auto()
    NON_TERMINAL = enum.auto()


_CODES = list(CheckResult)
_TERMINAL = set(_CODES[:7])


def unpack_state(idx: int) -> State:
    idx = int(idx)
    return (idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7)


def pack_state(state) -> int:
    k, p, v, a, t = (int(x) for x in state)
    return (((k * 3 + p) * 3 + v) * 3 + a) * 7 + t


class AbstractMdp:
    """Common part of pkg/mdp.py:80-203 (constructor keywords and defaults identical)."""

    def __init__(self, working_curriculum_step: int, f_ag: float, t_max: int, p_max: float = 4.5, *, w_p: float = -100.0,
                 w_v: float = -10.0, w_theta: float = -1.55, w_dur: float = -6.0, w_fail: float = -2.6, w_succ: float = 2.6,
                 n_theta: int = 3, v_max: float = 3.39411, a_max: float = 1.28, theta_max: float = float(np.deg2rad(21.37723)),
                 delta_theta: float = float(np.deg2rad(7.12574)), beta: float = 1 / 3, sigma_a: float = 0.416,
                 minimum_altitude: float = 0.1, mode: str = "reference", device: int = 0) -> None:
        if n_theta != 3:
            raise ValueError("only n_theta = 3 (7 angle bins, table shape (5,3,3,3,7,3)) is supported")
        if mode not in ("reference", "paper"):
            raise ValueError("mode must be 'reference' or 'paper'")
        self._working_curriculum_step = working_curriculum_step
        self._f_ag, self._t_max, self._p_max = f_ag, t_max, p_max
        self._flyzone_x = (-p_max, p_max)
        self._flyzone_y = (-p_max, p_max)
        self._flyzone_z = (0.0, p_max)
        self._w_p, self._w_v, self._w_theta, self._w_dur, self._w_fail, self._w_succ = w_p, w_v, w_theta, w_dur, w_fail, w_succ
        self._n_theta, self._theta_max, self._delta_theta = n_theta, theta_max, delta_theta
        self._v_max, self._a_max, self._beta, self._sigma_a = v_max, a_max, beta, sigma_a
        self._minimum_altitude = minimum_altitude
        self._discrete_angles = np.linspace(-theta_max, theta_max, (n_theta * 2) + 1)
        self._limits = Limits(working_curriculum_step)
        self._delta_t = 1 / self._f_ag
        self._device = device
        # float64 on the device: bit-identical to the reference's numpy arithmetic
        self._cfg = DqlConfig(working_curriculum_step=working_curriculum_step, dtype=F64, quirks=Q_REFERENCE if mode == "reference" else 0,
                              f_ag=f_ag, t_max=t_max, p_max=p_max, v_max=v_max, a_max=a_max, theta_max=theta_max, delta_theta=delta_theta,
                              beta=beta, sigma_a=sigma_a, minimum_altitude=minimum_altitude, w_p=w_p, w_v=w_v, w_theta=w_theta,
                              w_dur=w_dur, w_fail=w_fail, w_succ=w_succ)
        # shaping memory lives for the lifetime of the object (B9: reset() does not clear it in reference mode)
        self.current_shaping_value = RewardShapingValue()
        self.previous_shaping_value = RewardShapingValue()
        self._mode = mode
        self._info: Dict[str, Any] = {}
        self._step_count = 0
        self._check_result = CheckResult.NON_TERMINAL

    # -- helpers shared by the two flavours --
    def _axis_state(self):
        return np.array([[0.0], [0.0], [0.0], [0.0], [0.0], [0.0], [0.0], [float(_CODES.index(CheckResult.NON_TERMINAL))]])

    def _base_reset(self):
        self._info = {}
        self._step_count = 0
        self._check_result = CheckResult.NON_TERMINAL
        if self._mode == "paper":
            self.current_shaping_value = RewardShapingValue()
            self.previous_shaping_value = RewardShapingValue()

    def _terminal_info(self, obs: ContinuousObservation):
        r = self._check_result
        if r == CheckResult.TERMINAL_FLYZONE_X:
            self._info["Relative x"] = f"self._current_continuous_observation.rel_p_x={obs.rel_p_x}"
            self._info["Fly zone x"] = f"self._flyzone_x={self._flyzone_x}"
        elif r == CheckResult.TERMINAL_FLYZONE_Y:
            self._info["Relative y"] = f"self._current_continuous_observation.rel_p_y={obs.rel_p_y}"
            self._info["Fly zone y"] = f"self._flyzone_y={self._flyzone_y}"
        elif r == CheckResult.TERMINAL_MINIMUM_ALTITUDE:
            self._info["Relative z"] = f"self._current_continuous_observation.abs_p_z={obs.abs_p_z}"
            self._info["Fly zone z"] = f"self._flyzone_z={self._flyzone_z}"
        elif r == CheckResult.TERMINAL_FLYZONE_Z:
            self._info["Relative z"] = f"self._current_continuous_observation.rel_p_y={obs.rel_p_y}"  # sic (B12)
            self._info["Fly zone z"] = f"{self._flyzone_y}"
        elif r == CheckResult.TERMINAL_TIMEOUT:
            self._info["Timeout"] = f"self._t_max * self._f_ag ={self._t_max * self._f_ag}"

    def reward(self) -> float:
        return 0.0


class TrainingMdp(AbstractMdp):
    """pkg/mdp.py:206-569."""

    def __init__(self, working_curriculum_step: int, f_ag: float, t_max: int, p_max: float = 4.5, *, minimum_altitude: float = 0.2, **kw) -> None:
        super().__init__(working_curriculum_step, f_ag, t_max, p_max, minimum_altitude=minimum_altitude, **kw)
        self._current_continuous_observation = ContinuousObservation()
        self._current_discrete_state: Optional[State] = None
        self._previous_discrete_state: Optional[State] = None
        self._curriculum_check = 0
        self._cumulative_reward = 0
        self._current_continuous_action = Action(pitch=0, roll=0, yaw=0, v_z=-0.1)

    # device state block <-> attributes
    def _ms(self):
        s = self.current_shaping_value
        return np.array([[float(self._current_continuous_action.pitch)], [float(s.position)], [float(s.velocity)], [float(s.angle)],
                         [float(self._cumulative_reward)], [float(self._step_count)], [float(self._curriculum_check)],
                         [float(_CODES.index(self._check_result))]])

    def _obs(self):
        o = self._current_continuous_observation
        return np.array([[o.rel_p_x], [o.rel_p_y], [o.rel_v_x], [o.rel_a_x], [o.pitch], [o.abs_p_z], [1.0 if o.contact else 0.0]], dtype=np.float64)

    def _call(self, stages, action=2):
        prev = np.array([pack_state(self._previous_discrete_state) if self._previous_discrete_state else -1], dtype=np.int32)
        cur = np.array([pack_state(self._current_discrete_state) if self._current_discrete_state else -1], dtype=np.int32)
        return ops.mdp_transition(self._cfg, [action], self._obs(), self._ms(), prev, cur, stages=stages, device=self._device)

    def discrete_state(self, current_continuous_observation: ContinuousObservation) -> State:
        self._previous_discrete_state = self._current_discrete_state
        self._current_continuous_observation = current_continuous_observation
        _, idx, _, _ = self._call(ops.MDP_DISCRETISE)
        if idx[0] < 0:
            raise ValueError("Unexpected discretization case")
        self._current_discrete_state = unpack_state(idx[0])
        return self._current_discrete_state

    def check(self):
        if not self._current_discrete_state:
            raise ValueError("Cannot check an empty state\nYou must call `discrete_state` before calling check.")
        ms, _, _, _ = self._call(ops.MDP_CHECK)
        self._step_count = int(ms[5, 0])
        self._curriculum_check = int(ms[6, 0])
        self._check_result = _CODES[int(ms[7, 0])]
        self._terminal_info(self._current_continuous_observation)
        if self._check_result in _TERMINAL:
            self._info["Termination condition"] = self._check_result.value
            self._info["Number of steps"] = self._step_count
            self._info["Cumulative reward"] = self._cumulative_reward
            self._info["Mean reward"] = self._cumulative_reward / self._step_count
        return self._info

    def reward(self) -> float:
        if not self._previous_discrete_state:
            raise ValueError("Previous state missing.\nYou must call `reset` and `discrete_state`and then `step`before calling check.")
        if not self._current_discrete_state:
            raise ValueError("Cannot check an empty state.\nYou must call `discrete_state` before calling check.")
        self.previous_shaping_value = RewardShapingValue(self.current_shaping_value.position, self.current_shaping_value.velocity,
                                                         self.current_shaping_value.angle)
        ms, _, rew, _ = self._call(ops.MDP_REWARD)
        self.current_shaping_value = RewardShapingValue(float(ms[1, 0]), float(ms[2, 0]), float(ms[3, 0]))
        self._cumulative_reward = float(ms[4, 0])
        return float(rew[0])

    def continuous_action(self, action_x: int, action_y: int = 2):
        if action_y != 2:
            raise ValueError("Cannot move in the y direction while training")
        if action_x in (0, 1):
            ms, _, _, _ = self._call(ops.MDP_ACTION, action=int(action_x))
            self._current_continuous_action.pitch = float(ms[0, 0])
        return self._current_continuous_action

    def reset(self):
        self._base_reset()
        self._current_continuous_observation = ContinuousObservation()
        self._current_discrete_state = None
        self._previous_discrete_state = None
        self._curriculum_check = 0
        self._cumulative_reward = 0
        self._current_continuous_action = Action(pitch=0, roll=0, yaw=0, v_z=-0.1)


class SimulationMdp(AbstractMdp):
    """pkg/mdp.py:572-886: two 1-D discretisations (x with pitch, y with roll), terminal checks without goal logic."""

    def __init__(self, working_curriculum_step: int, f_ag: float, t_max: int, *, p_max: float = 4.5, minimum_altitude: float = 0.2, **kw) -> None:
        super().__init__(working_curriculum_step, f_ag, t_max, p_max, minimum_altitude=minimum_altitude, **kw)
        self._current_continuous_observation = ContinuousObservation()
        self._current_discrete_state_x: Optional[State] = None
        self._previous_discrete_state_x: Optional[State] = None
        self._current_discrete_state_y: Optional[State] = None
        self._previous_discrete_state_y: Optional[State] = None
        self._current_continuous_action = Action(pitch=0, roll=0, yaw=0, v_z=-0.4)

    def _discretise_axis(self, axis: str) -> State:
        o = self._current_continuous_observation
        if axis == "x":
            p, v, a, ang = o.rel_p_x, o.rel_v_x, o.rel_a_x, o.pitch
        else:
            p, v, a, ang = o.rel_p_y, o.rel_v_y, o.rel_a_y, o.roll
        idx = ops.discretise(self._cfg, [p], [v], [a], [ang], device=self._device)
        if idx[0] < 0:
            raise ValueError("Unexpected discretization case")
        return unpack_state(idx[0])

    def discrete_state(self, current_continuous_observation: ContinuousObservation) -> Tuple[State, State]:
        self._previous_discrete_state_x = self._current_discrete_state_x
        self._previous_discrete_state_y = self._current_discrete_state_y
        self._current_continuous_observation = current_continuous_observation
        return self.discrete_state_x(), self.discrete_state_y()

    def discrete_state_x(self) -> State:
        self._current_discrete_state_x = self._discretise_axis("x")
        return self._current_discrete_state_x

    def discrete_state_y(self) -> State:
        self._current_discrete_state_y = self._discretise_axis("y")
        return self._current_discrete_state_y

    def check(self):
        if not self._current_discrete_state_x or not self._current_discrete_state_y:
            raise ValueError("Cannot check an empty state\nYou must call `discrete_state` before calling check.")
        o = self._current_continuous_observation
        obs = np.array([[o.rel_p_x], [o.rel_p_y], [o.rel_v_x], [o.rel_a_x], [o.pitch], [o.abs_p_z], [1.0 if o.contact else 0.0]], dtype=np.float64)
        ms = self._axis_state()
        ms[5, 0] = self._step_count
        ms[7, 0] = _CODES.index(self._check_result)
        cur = np.array([pack_state(self._current_discrete_state_x)], dtype=np.int32)
        ms, _, _, _ = ops.mdp_transition(self._cfg, [2], obs, ms, np.array([-1], dtype=np.int32), cur,
                                         stages=ops.MDP_CHECK | ops.MDP_SIMULATION, device=self._device)
        self._step_count = int(ms[5, 0])
        self._check_result = _CODES[int(ms[7, 0])]
        self._terminal_info(o)
        if self._check_result in _TERMINAL:
            self._info["Termination condition"] = self._check_result.value
            self._info["Number of steps"] = self._step_count
        return self._info

    def continuous_action(self, action_x: int, action_y: int):
        if action_x in (0, 1):
            ms = self._axis_state()
            ms[0, 0] = float(self._current_continuous_action.pitch)
            ms, _, _, _ = ops.mdp_transition(self._cfg, [int(action_x)], np.zeros((7, 1)), ms, np.array([-1], dtype=np.int32),
                                             stages=ops.MDP_ACTION, device=self._device)
            self._current_continuous_action.pitch = float(ms[0, 0])
        # the roll action is dead code in the reference (`if False and ...`, pkg/mdp.py:863-876, B16)
        return self._current_continuous_action

    def reset(self):
        self._base_reset()
        self._current_continuous_observation = ContinuousObservation()
        self._current_discrete_state_x = None
        self._previous_discrete_state_x = None
        self._current_discrete_state_y = None
        self._previous_discrete_state_y = None
        self._current_continuous_action = Action(pitch=0.0, roll=0.0, yaw=0, v_z=-0.4)
