"""Drop-in mirror of the reference's `trainer.py` (pkg/trainer.py:19-303): same constructor keywords, schedules
(`alpha`, `exploration_rate`, `transfer_learning_ratio`), promotion rule, checkpoint file names and directory layout.
`curriculum_training()` runs the reference's loop (pkg/trainer.py:169-245) for `n_envs` environments at once on the
GPU: `Engine.train_steps` is the inner `while not done` body (guess, env.step, agent.update) of every env, and the
host only evaluates the episode-indexed schedules between chunks of agent periods.

Build-specific keywords (not in the reference): n_envs, device, dtype, mode, chunk_steps, checkpoint_every, quiet.
Trainer state is saved as JSON (never pickle); the reference's resume path is broken (B12), this one works."""
from __future__ import annotations

import csv
import json
import math
import time
from collections import deque
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

from .config import DqlConfig, F32, Q_PAPER, Q_REFERENCE
from .double_q_learning import ASSETS_PATH, DoubleQLearningAgent, StateAction
from .engine import Engine

_TS = r"%d-%m-%Y %H:%M:%S"


class Trainer:
    def __init__(self, curriculum_steps: int = 5, double_q_learning_agent: Optional[DoubleQLearningAgent] = None,
                 successive_successful_episodes: int = 100, success_rate: float = 0.96, max_num_episodes: int = 50000,
                 initial_curriculum_step: int = 0, seed: int = 42, save_path=None, *, alpha_min: float = 0.02949, omega: float = 0.51,
                 gamma: float = 0.99, scale_modification_value=(0.8172650252856599, 0.8211253690681617, 0.8257273369742982, 0.8311571820651724),
                 t_max: int = 20, z_init: float = 4.0, f_ag: float = 22.92, p_max: float = 4.5,
                 n_envs: int = 4096, device: int = 0, dtype: int = F32, mode: str = "reference", chunk_steps: int = 64,
                 checkpoint_every: int = 50, max_steps_per_level: Optional[int] = None, quiet: bool = True,
                 fold_per_step: int = 0, eps_floor: float = 0.0) -> None:
        np.random.seed(seed)
        if mode not in ("reference", "paper"):
            raise ValueError("mode must be 'reference' or 'paper'")
        if not double_q_learning_agent:
            double_q_learning_agent = DoubleQLearningAgent(curriculum_steps)
        self._double_q_learning_agent = double_q_learning_agent
        self._curriculum_steps = self._double_q_learning_agent.curriculum_steps
        self._alpha_min, self._omega, self._gamma = alpha_min, omega, gamma
        self._scale_modification_value = list(scale_modification_value)
        self._successive_successful_episodes = successive_successful_episodes
        self._success_rate = success_rate
        self._alpha = self._alpha_min
        self._exploration_rate = 0.0
        self._z_init, self._t_max, self._f_ag, self._p_max = z_init, t_max, f_ag, p_max
        self._max_num_episodes = max_num_episodes
        self._save_path: Path = Path(save_path) if save_path is not None else ASSETS_PATH / datetime.now().strftime(_TS)
        self._seed = seed
        self._current_episode = 0
        self._working_curriculum_step = initial_curriculum_step
        self._curriculum_episode_count = 0
        self._successes = deque([], maxlen=successive_successful_episodes)
        # build-specific
        self._n_envs, self._device, self._dtype, self._mode = int(n_envs), device, dtype, mode
        self._chunk_steps, self._checkpoint_every, self._quiet = int(chunk_steps), int(checkpoint_every), quiet
        self._max_steps_per_level = max_steps_per_level
        self._fold_per_step, self._eps_floor = int(fold_per_step), float(eps_floor)
        self.history = []  # one record per finished curriculum level
        self._engine: Optional[Engine] = None

    # ---- schedules: pkg/trainer.py:88-138 ----
    def alpha(self, current_state_action: StateAction):
        counter = self._double_q_learning_agent.state_action_counter[current_state_action]
        if counter == 0:
            self._alpha = self._alpha_min
        else:
            self._alpha = float(np.max([np.float_power(1 / (counter), self._omega), self._alpha_min]))
        if math.isnan(self._alpha):
            raise ValueError(f"Leaning rate cannot be NaN, {counter}, {self._omega}, {self._alpha_min}")
        return self._alpha

    def exploration_rate(self, current_episode: int, current_curriculum_step: int):
        if current_curriculum_step > 0:
            self._exploration_rate = 0.0
        elif 0 <= current_episode <= 800:
            self._exploration_rate = 1.0
        else:
            self._exploration_rate = max(1 + (0.01 - 1) * (current_episode - 800) / (2000 - 800), 0.01)
        return self._exploration_rate

    def transfer_learning_ratio(self, curriculum_step: int) -> float:
        if curriculum_step < 1:
            return 1.0
        elif curriculum_step < (len(self._scale_modification_value) + 1):
            return self._scale_modification_value[curriculum_step - 1]
        raise ValueError(f"Transfer learning can be done up to he 5th curiculum_step, {curriculum_step} is invalid")

    # ---- checkpoints: pkg/trainer.py:140-167 (same .npy names, run dir + copy one level up) ----
    def _state_dict(self):
        return {"curriculum_steps": self._curriculum_steps, "working_curriculum_step": self._working_curriculum_step,
                "current_episode": self._current_episode, "curriculum_episode_count": self._curriculum_episode_count,
                "seed": self._seed, "alpha_min": self._alpha_min, "omega": self._omega, "gamma": self._gamma,
                "scale_modification_value": self._scale_modification_value, "t_max": self._t_max, "z_init": self._z_init,
                "f_ag": self._f_ag, "p_max": self._p_max, "success_rate": self._success_rate,
                "successive_successful_episodes": self._successive_successful_episodes, "max_num_episodes": self._max_num_episodes,
                "n_envs": self._n_envs, "mode": self._mode, "history": self.history}

    def save(self) -> None:
        self._pull_tables()
        self._save_path.mkdir(parents=True, exist_ok=True)
        with open(self._save_path / "trainer.json", "w") as f:
            json.dump(self._state_dict(), f, indent=1)
        self._double_q_learning_agent.save(self._save_path)
        self._double_q_learning_agent.save(self._save_path / "..")

    @staticmethod
    def load(assets_path: Path = ASSETS_PATH, **kw) -> "Trainer":
        assets_path = Path(assets_path)
        runs = []
        for p in assets_path.iterdir():
            try:
                runs.append((datetime.strptime(p.name, _TS), p))
            except ValueError:
                continue
        if not runs:
            raise FileNotFoundError(f"no run directory named like '{_TS}' under {assets_path}")
        run = max(runs)[1]
        with open(run / "trainer.json") as f:
            st = json.load(f)
        agent = DoubleQLearningAgent.load(run)
        tr = Trainer(curriculum_steps=st["curriculum_steps"], double_q_learning_agent=agent, initial_curriculum_step=st["working_curriculum_step"],
                     seed=st["seed"], save_path=run, alpha_min=st["alpha_min"], omega=st["omega"], gamma=st["gamma"],
                     scale_modification_value=st["scale_modification_value"], t_max=st["t_max"], z_init=st["z_init"], f_ag=st["f_ag"],
                     p_max=st["p_max"], success_rate=st["success_rate"], successive_successful_episodes=st["successive_successful_episodes"],
                     max_num_episodes=st["max_num_episodes"], n_envs=kw.pop("n_envs", st["n_envs"]), mode=st["mode"], **kw)
        tr._current_episode, tr._curriculum_episode_count, tr.history = st["current_episode"], st["curriculum_episode_count"], st["history"]
        return tr

    # ---- device plumbing ----
    def _config(self, level: int) -> DqlConfig:
        return DqlConfig(working_curriculum_step=level, dtype=self._dtype, quirks=Q_REFERENCE if self._mode == "reference" else Q_PAPER,
                         t_max=self._t_max, z_init=self._z_init, f_ag=self._f_ag, p_max=self._p_max, init_sigma=self._p_max / 3,
                         gamma=self._gamma, alpha_min=self._alpha_min, alpha_omega=self._omega, fold_per_step=self._fold_per_step)

    def _push_tables(self):
        a = self._double_q_learning_agent
        qa, qb, cnt = a._padded()
        self._engine.set_tables(qa, qb, cnt)

    def _pull_tables(self):
        if self._engine is None:
            return
        qa, qb, cnt = self._engine.get_tables()
        self._double_q_learning_agent._unpad(qa.reshape(-1), qb.reshape(-1), cnt.reshape(-1))

    # ---- pkg/trainer.py:169-245 ----
    def curriculum_training(self):
        t_start = time.perf_counter()
        cfg = self._config(self._working_curriculum_step)
        self._engine = Engine(cfg, self._n_envs, seed=self._seed, device=self._device)
        self._push_tables()
        eng = self._engine
        for self._working_curriculum_step in range(self._working_curriculum_step, self._curriculum_steps):
            k = self._working_curriculum_step
            if self._mode == "paper" and k >= 1:
                eng.transfer(k, self.transfer_learning_ratio(k))  # Eq. 31 as intended: the new level starts from the previous one
            eng.set_curriculum(k)  # "Create a new environment to update limits" (:175-183)
            s_prev = eng.stats()
            t_level = time.perf_counter()
            window = deque()  # (episodes, goal-state successes) per chunk, trimmed to the most recent >= 100 episodes
            episodes = 0
            steps = 0
            promoted = False
            info: Dict[str, Any] = {}
            chunk_i = 0
            while episodes < self._max_num_episodes:
                eps = max(self.exploration_rate(episodes, k), self._eps_floor)  # eps_floor = 0 is the reference schedule
                eng.train_steps(self._chunk_steps, eps)
                steps += self._chunk_steps
                s = eng.stats()
                new_eps = s["episodes"] - s_prev["episodes"]
                new_ok = s["by_code"]["TERMINAL_SUCCESS"] - s_prev["by_code"]["TERMINAL_SUCCESS"]  # "Goal state reached" only (B17)
                new_dec = s["decisions"] - s_prev["decisions"]
                new_rew = s["reward_sum"] - s_prev["reward_sum"]
                s_prev = s
                episodes += new_eps
                self._current_episode = episodes
                self._curriculum_episode_count += new_eps
                if new_eps:
                    window.append((new_eps, new_ok))
                    while sum(e for e, _ in window) - window[0][0] >= self._successive_successful_episodes:
                        window.popleft()
                w_eps = sum(e for e, _ in window)
                w_ok = sum(o for _, o in window)
                # the reference divides by the deque length limit (100) also while the deque is filling (:222-224)
                rate = w_ok / max(w_eps, self._successive_successful_episodes)
                info = {"Curent episode": episodes, "Remaining episodes": self._max_num_episodes - episodes + 1, "Exploration rate": eps,
                        "Learning rate": self._alpha, "Success rate": rate, "Mean reward": new_rew / max(1, new_dec), "Agent periods": steps,
                        "Curriculum step": k}
                chunk_i += 1
                if chunk_i % self._checkpoint_every == 0:
                    self.save()
                self.log(info)
                if rate > self._success_rate:
                    self._successes = deque([], maxlen=self._successive_successful_episodes)
                    promoted = True
                    break
                if self._max_steps_per_level is not None and steps >= self._max_steps_per_level:
                    break
            self.history.append({"level": k, "promoted": promoted, "episodes": episodes, "agent_periods": steps, "success_rate": info.get("Success rate"),
                                 "wall_s": time.perf_counter() - t_level, "wall_since_start_s": time.perf_counter() - t_start})
            if self._mode == "reference":
                # transfer AFTER finishing level k: Q[k] = Q[k-1] * ratio, k = 0 wraps (B6, pkg/trainer.py:237-243)
                eng.transfer(k, self.transfer_learning_ratio(k))
            self.save()
            if not promoted:
                break
        return self.history

    # ---- pkg/trainer.py:247-303: scalar log with the reference's tag names (CSV instead of one TensorBoard file per episode) ----
    def log(self, info: Dict[str, Any], clean=False):
        path = self._save_path / "logs"
        path.mkdir(parents=True, exist_ok=True)
        f = path / "scalars.csv"
        new = not f.exists()
        with open(f, "a", newline="") as fh:
            w = csv.writer(fh)
            if new:
                w.writerow(["Curriculum step", "Curent episode", "Agent periods", "Success Rate", "Exploration Rate", "Learning Rate", "Mean reward"])
            w.writerow([info.get("Curriculum step"), info.get("Curent episode"), info.get("Agent periods"), info.get("Success rate"),
                        info.get("Exploration rate"), info.get("Learning rate"), info.get("Mean reward")])
        if not self._quiet:
            print(" | ".join(f"{k}: {v}" for k, v in info.items()), flush=True)
