"""Drop-in mirror of the reference's `landing_simulation_env.py` (pkg/landing_simulation_env.py): the gym-style
`TrainingLandingEnv` / `SimulationLandingEnv` for ONE environment with the reference's signatures, and the vectorised
`VecLandingEnv` the GPU is built for.  Gazebo + the five ROS nodes are replaced by the fused HIP step kernel
(`Engine.step`); there is no gym dependency (the reference only uses `gym.Env` as a base class and `gym.make`)."""
from __future__ import annotations

from typing import Any, Dict, Tuple

import numpy as np

from .config import CHECK_NAMES, DqlConfig, F64, Q_PAPER, Q_REFERENCE
from .engine import Engine
from .mdp import CheckResult, unpack_state

_CODES = list(CheckResult)


class VecLandingEnv:
    """N landing environments stepped together on one GPU.

    reset()            -> states int32[N,5]
    step(actions[N])   -> (states int32[N,5], rewards float64[N], dones bool[N], info dict of arrays)
    Finished envs are reset by the NEXT step call (one agent period of simulation, as the reference's reset() runs);
    for those envs that step returns the first state of the new episode with reward 0 and info["was_reset"] set.
    """

    def __init__(self, n_envs: int, initial_curriculum_step: int = 0, *, t_max: int = 20, f_ag: float = 22.92, p_max: float = 4.5,
                 z_init: float = 2.0, seed: int = 42, device: int = 0, dtype: int = F64, mode: str = "reference", config: DqlConfig = None):
        if config is None:
            config = DqlConfig(working_curriculum_step=initial_curriculum_step, t_max=t_max, f_ag=f_ag, p_max=p_max, z_init=z_init,
                               init_sigma=p_max / 3, dtype=dtype, quirks=Q_REFERENCE if mode == "reference" else Q_PAPER)
        self.cfg = config
        self.n = int(n_envs)
        self.engine = Engine(config, self.n, seed=seed, device=device)
        self._working_curriculum_step = config.working_curriculum_step

    @staticmethod
    def _tuples(idx):
        idx = np.asarray(idx, dtype=np.int64)
        return np.stack([idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7], axis=-1).astype(np.int32)

    def reset(self, mask=None):
        self.engine.reset(mask)
        self.engine.step(np.full(self.n, 2, dtype=np.uint8))  # the reset period: placement + one agent period
        return self._tuples(self.engine.states())

    def step(self, actions):
        self.engine.step(actions)
        o = self.engine.step_outputs()  # one device round trip (no full state download)
        info = {"check_code": o["code"].astype(np.int32), "step_count": o["step_count"], "cumulative_reward": o["cumulative_reward"], "was_reset": o["was_reset"] != 0}
        return self._tuples(o["idx_x"]), o["reward"], o["done"] != 0, info

    def close(self):
        self.engine.close()


class _SingleEnv:
    """Shared N = 1 plumbing of the two reference-shaped environments."""

    def __init__(self, cfg: DqlConfig, seed: int, device: int):
        self._vec = VecLandingEnv(1, config=cfg, seed=seed, device=device)
        self._info: Dict[str, Any] = {}
        self._act = np.zeros(1, dtype=np.uint8)  # the one action byte handed to dql_step, allocated once

    def _step1(self, action: int):
        """one env, one step: dql_step + dql_step_outputs on preallocated buffers (the per-step cost of this API is host overhead)"""
        eng = self._vec.engine
        self._act[0] = action
        eng.step_raw(self._act)
        o = eng.step_outputs_view()
        idx = int(o["idx_x"][0])
        state = (idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7)
        return state, float(o["reward"][0]), o

    def close(self):
        self._vec.close()

    def _info_for(self, info, reward=None):
        code = _CODES[int(info["check_code"][0])]
        if int(info["check_code"][0]) <= 6:
            self._info["Termination condition"] = code.value
            self._info["Number of steps"] = int(info["step_count"][0])
            if reward is not None:
                self._info["Cumulative reward"] = float(info["cumulative_reward"][0]) - float(reward)  # check() runs before reward()
                self._info["Mean reward"] = self._info["Cumulative reward"] / max(1, int(info["step_count"][0]))
        return self._info


class TrainingLandingEnv(_SingleEnv):
    """pkg/landing_simulation_env.py:142-282."""

    def __init__(self, initial_curriculum_step: int = 0, *, t_max: int = 20, f_ag: float = 22.92, p_max: float = 4.5, z_init: float = 2.0,
                 seed: int = 42, device: int = 0, mode: str = "reference", dtype: int = F64):
        """`dtype` (build-specific): F64 (default) flies the reference's expressions operation by operation — what golden G13 pins bit for bit;
        F32 flies the float32 step every throughput figure is measured on (within 1e-5 relative of float64 over an agent period), whose
        one-env kernel is 2.5x shorter: the choice for a host loop that wants speed, not bit-identity with the reference's Python."""
        cfg = DqlConfig(working_curriculum_step=initial_curriculum_step, t_max=t_max, f_ag=f_ag, p_max=p_max, z_init=z_init, init_sigma=p_max / 3,
                        dtype=dtype, quirks=Q_REFERENCE if mode == "reference" else Q_PAPER)
        super().__init__(cfg, seed, device)
        self._working_curriculum_step = initial_curriculum_step

    def reset(self) -> Tuple[int, int, int, int, int]:
        self._info = {}
        s = self._vec.reset()
        return tuple(int(x) for x in s[0])

    def step(self, action_x: int, action_y: int = 2):
        if action_y != 2:
            raise ValueError("Cannot move in the y direction while training")
        if action_x not in (0, 1, 2):
            raise ValueError("action_x must be 0 (increase), 1 (decrease) or 2 (hold)")
        state, r, o = self._step1(action_x)
        out = self._info_for({"check_code": o["code"], "step_count": o["step_count"], "cumulative_reward": o["cumulative_reward"]}, reward=r)
        out["Current reward"] = r
        return state, r, "Termination condition" in out.keys(), out


class SimulationLandingEnv(_SingleEnv):
    """pkg/landing_simulation_env.py:285-428: greedy evaluation flavour (level 4, v_z -0.4 m/s, uniform start,
    z_init 4).  The y axis is never flown in the reference (B16): the y state is discretised from the same
    observation, the roll set-point stays 0."""

    def __init__(self, initial_curriculum_step: int = 4, *, t_max: int = 20, f_ag: float = 22.92, p_max: float = 4.5, z_init: float = 4,
                 seed: int = 42, device: int = 0):
        from .config import simulation_config
        cfg = simulation_config(working_curriculum_step=initial_curriculum_step, t_max=t_max, f_ag=f_ag, p_max=p_max, z_init=z_init, dtype=F64)
        super().__init__(cfg, seed, device)
        from .mdp import SimulationMdp
        self._mdp_y = SimulationMdp(initial_curriculum_step, f_ag, t_max, p_max=p_max, device=device)

    def _state_y(self):
        from . import ops
        reals, _ = self._vec.engine.get_fields()
        nm = self._vec.engine.field_names()
        g = lambda k: reals[nm.index(k)]
        qw, qx, qy, qz = g("qw")[0], g("qx")[0], g("qy")[0], g("qz")[0]
        roll = float(np.arctan2(2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)))  # euler_from_quaternion, axes sxyz
        idx = ops.discretise(self._vec.cfg, g("obs_p_y"), g("obs_v_y"), g("obs_a_y"), [roll])
        return unpack_state(idx[0])

    def reset(self):
        self._info = {}
        s = self._vec.reset()
        return tuple(int(x) for x in s[0]), self._state_y()

    def step(self, action_x: int, action_y: int):
        if action_x not in (0, 1, 2) or action_y not in (0, 1, 2):
            raise ValueError("actions must be 0 (increase), 1 (decrease) or 2 (hold)")
        s, r, d, info = self._vec.step(np.array([action_x], dtype=np.uint8))
        code = int(info["check_code"][0])
        # SimulationMdp.check has no goal logic (pkg/mdp.py:784-845): only the terminal failure / contact codes end it
        if code <= 6 and code != 1:
            self._info["Termination condition"] = _CODES[code].value
            self._info["Number of steps"] = int(info["step_count"][0])
        return tuple(int(x) for x in s[0]), self._state_y(), "Termination condition" in self._info.keys(), self._info
