"""Drop-in mirror of the reference's `trainer.py` (pkg/trainer.py:19-303): same constructor keywords, schedules
(`alpha`, `exploration_rate`, `transfer_learning_ratio`), promotion rule, checkpoint file names and directory layout.
`curriculum_training()` runs the reference's loop (pkg/trainer.py:169-245) for `n_envs` environments at once on the
GPU: `Engine.train_steps` is the inner `while not done` body (guess, env.step, agent.update) of every env, and the
host only evaluates the episode-indexed schedules between chunks of agent periods.

`mode`: "reference" reproduces the reference's behaviour including its quirks (only Q_table_a is
updated and values itself, transfer after a level with the k = 0 wrap, ...); "paper" is what the code was written to do:
Double Q-learning (a coin picks the table to update, the other one values its greedy action), quirks B3 / B7 / B8 / B9 / B19
off, transfer before the new level.

Promotion (`promotion_rule`): "ordered" is the reference's rule itself — a deque of the last 100 episodes, checked after
every episode — on the episodes of the first `judge_envs` envs, generation by generation: all first episodes of the level
in env order, then all second ones, ... (promotion.py explains why neither completion nor start order will do; the engine's
episode log supplies what is needed); "aggregate" is the stricter large-sample form
(success rate over all episodes of the most recent chunks covering >= 100 episodes).  Like
the reference (:187, the `for` over `max_num_episodes` simply ends), a level whose episode budget runs out without a
promotion still hands over to the next level.

Multi-GPU (BASELINE config 4): launched as one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in
the environment, e.g. by torch.distributed.run or `bench.py --gpus N`) every rank runs this same loop on its shard of the
`n_envs` global envs; tables synchronise every `sync_period` agent periods through dist.ShardedRunner (RCCL all-reduce issued
by libdql_hip.so, comm.py), the per-chunk counters are all-reduced and the judged envs' episode logs all-gathered (rank order
= global env order), so every rank takes the same promotion decisions.  What is invariant: for a given `sync_period` the run
does not depend on the number of ranks — one rank included: `sync_period=S` on a single GPU follows the same windowed table
schedule (exchange = identity) as 8 GPUs with `sync_period=S`, bit for bit for S <= 2 and per schedule beyond.
`sync_period=None` (default) means: no windows on one rank (tables act one period late), 2 on several.

Build-specific keywords (not in the reference): n_envs, device, dtype, mode, chunk_steps, checkpoint_every, quiet,
promotion_rule, judge_envs, sync_period, max_steps_per_level, eps_floor, quirks (override of the mode's quirk set, include/dql.h DQL_Q_*), eps_episode_scale (the reference's
exploration schedule counts episodes of ONE env: 800 random episodes, 1 200 decaying; N envs finish that many in their first
generation, so `eps_episode_scale = s` reads the schedule at episodes / s), checkpoint_env_state, periods_per_launch (agent periods
per kernel launch, engine option of the same name: the tables the envs act on are refreshed once per launch; 1 = every period),
eps_tail (None = the reference: level 0 keeps exploring at 0.01 once the schedule's decay has ended, pkg/trainer.py:121-131; a number
replaces that tail.  Measured on 32 768 envs, tools/exp_level0.py -> profiles/r3_level0_eps_tail.jsonl: tables LEARNT from the 1 % of
exploratory transitions fly 58 % of level 0's episodes into the goal state and 41 % out of the fly zone, whether they then ACT with
eps 0.01 or 0; the same tables learning on with eps 0 reach 99.4 % within 250 agent periods.  The paper's text ends the schedule at
episode 2000; the later levels run at eps 0 in both).
`max_num_episodes=None` (default) is the reference's 50 000 per level, but at least 384 per env: the reference's figure is
sized for one env, and a level has to last until the judged envs have flown a few hundred episodes each.
`judge_envs` (default 1): whose episodes feed the promotion deque.  1 is the reference's own situation — ONE env's episodes in the
order it flies them; every further judged env adds windows per unit of training, so levels hand over earlier and less trained
(measured, profiles/r2_curriculum_reference_counter_sweep*.jsonl: 4 096 judged envs promote after 3-60 episodes per env and land
70 % of the final policy's approaches, 1 judged env after 100-400 episodes per env and lands 90 %).
`fold_per_step` (default 1): how a launch's m visits of a table cell move its value.  0: as m sequential visits (the
contraction over alpha(c) .. alpha(c+m-1)); 1: one learning-rate step towards the launch's mean target.  For one env both
are the reference's rule (m <= 1).  With thousands of envs a cell collects hundreds of visits per launch and the
sequential form has no memory left — every launch's batch mean replaces the value, and the policies that come out lose the
platform in 8–25 % of the episodes (profiles/r1_stage4_many_envs_attempts.jsonl); the per-launch step averages over ~1/alpha
launches the way the reference averages over visits, needs an episode budget of ~64 per env and level, and lands at the
reference tables' touchdown rate with fly-zone exits at the infeasible-start floor (profiles/r1_stage4_per_step_fold.jsonl).

Checkpoints: trainer state as JSON (never pickle) + the three `.npy` tables in the reference's layout + (with
`checkpoint_env_state`) every env's simulator state, the engine's period index and the promotion bookkeeping, so that
`Trainer.load(dir).curriculum_training()` continues the interrupted level where it stopped — same history and tables as the
uninterrupted run with the same checkpoint schedule (a checkpoint is a table barrier: pending updates are folded and the acting
tables refreshed, in both).  The reference's resume path is broken (B12)."""
from __future__ import annotations

import csv
import json
import math
import os
import time
import warnings
from collections import deque
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

from .comm import RcclComm
from .config import CHECK_NAMES, DqlConfig, F32, Q_PAPER, Q_REFERENCE
from .dist import LocalWindowReducer, ShardedRunner, shard_range
from .double_q_learning import ASSETS_PATH, DoubleQLearningAgent, StateAction
from .engine import Engine
from .promotion import EpisodeOrder, PromotionWindow
from .state_layout import STATE_LAYOUT_VERSION, convert_env_state

_TS = r"%d-%m-%Y %H:%M:%S"
_TERMINAL = CHECK_NAMES[:7]
# scalar tags of the reference's SummaryWriter (pkg/trainer.py:252-279), one CSV row per chunk of agent periods
LOG_COLUMNS = (["Curriculum step", "Curriculum episode count", "Curent episode", "Agent periods", "Episode/Success Rate", "Episode/Cumulative Reward",
                "Episode/Exploration Rate", "Episode/Learning Rate", "Episode/Mean reward"]
               + [f"Episode/Termination Condition/{c}" for c in _TERMINAL])

# constructor keywords that are not in the reference; saved in trainer.json and restored by load()
_BUILD_KEYS = ("n_envs", "device", "dtype", "mode", "chunk_steps", "checkpoint_every", "max_steps_per_level", "quiet", "fold_per_step", "eps_floor",
               "promotion_rule", "sync_period", "judge_envs", "eps_episode_scale", "quirks", "checkpoint_env_state", "periods_per_launch", "eps_tail", "eps_tail_after",
               "population_gate", "env_kw", "restart_after", "transfer_counts", "step_back_after", "max_step_backs")


class Trainer:
    def __init__(self, curriculum_steps: int = 5, double_q_learning_agent: Optional[DoubleQLearningAgent] = None,
                 successive_successful_episodes: int = 100, success_rate: float = 0.96, max_num_episodes: Optional[int] = None,
                 initial_curriculum_step: int = 0, seed: int = 42, save_path=None, *, alpha_min: float = 0.02949, omega: float = 0.51,
                 gamma: float = 0.99, scale_modification_value=(0.8172650252856599, 0.8211253690681617, 0.8257273369742982, 0.8311571820651724),
                 t_max: int = 20, z_init: float = 4.0, f_ag: float = 22.92, p_max: float = 4.5,
                 n_envs: int = 4096, device: Optional[int] = None, dtype: int = F32, mode: str = "reference", chunk_steps: int = 64,
                 checkpoint_every: int = 50, max_steps_per_level: Optional[int] = None, quiet: bool = True,
                 fold_per_step: int = 1, eps_floor: float = 0.0, promotion_rule: str = "ordered", sync_period: Optional[int] = None,
                 judge_envs: Optional[int] = 1, eps_episode_scale: float = 1.0, quirks: Optional[int] = None, checkpoint_env_state: bool = True,
                 periods_per_launch: int = 1, eps_tail: Optional[float] = None, eps_tail_after: float = 0.0, population_gate: Optional[float] = None,
                 env_kw: Optional[Dict[str, Any]] = None, restart_after: Optional[float] = None, transfer_counts: float = 0.0, step_back_after: Optional[int] = None, max_step_backs: int = 3,
                 comm=None, reducer_factory=None) -> None:
        np.random.seed(seed)
        if mode not in ("reference", "paper"):
            raise ValueError("mode must be 'reference' or 'paper'")
        if promotion_rule not in ("ordered", "aggregate"):
            raise ValueError("promotion_rule must be 'ordered' or 'aggregate'")
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
        self._save_path: Path = Path(save_path) if save_path is not None else ASSETS_PATH / datetime.now().strftime(_TS)
        self._seed = seed
        self._current_episode = 0
        self._working_curriculum_step = initial_curriculum_step
        self._curriculum_episode_count = 0
        self._successes = deque([], maxlen=successive_successful_episodes)
        # build-specific
        self._n_envs, self._dtype, self._mode = int(n_envs), dtype, mode
        # the reference's 50 000 episodes per level are sized for ONE env; with N envs the level has to last until the judged env(s)
        # have flown a few hundred episodes (the deque wants 100 of them), i.e. a few hundred episodes PER env
        self._max_num_episodes = max(50000, 384 * self._n_envs) if max_num_episodes is None else int(max_num_episodes)
        self._chunk_steps, self._checkpoint_every, self._quiet = int(chunk_steps), int(checkpoint_every), quiet
        self._max_steps_per_level = max_steps_per_level
        self._fold_per_step, self._eps_floor = int(fold_per_step), float(eps_floor)
        self._promotion_rule = promotion_rule
        self._sync_period = None if sync_period is None else int(sync_period)
        self._eps_episode_scale = float(eps_episode_scale)
        self._quirks = (Q_REFERENCE if mode == "reference" else Q_PAPER) if quirks is None else int(quirks)
        self._judge_envs_arg = judge_envs
        self._judge_envs = self._n_envs if judge_envs is None else max(1, min(int(judge_envs), self._n_envs))
        self._checkpoint_env_state = bool(checkpoint_env_state)
        self._periods_per_launch = int(periods_per_launch)
        self._eps_tail = None if eps_tail is None else float(eps_tail)
        self._eps_tail_after = float(eps_tail_after)
        # population_gate: with N envs at once the reference's deque sees the episodes of the judged env(s) only — a 97 / 100 window of one or two
        # envs can pass while the population still fails 10 % of its episodes.  The gate keeps the reference's rule as it is and adds a second,
        # necessary condition: the success rate of ALL envs over the chunks holding the most recent >= 100 episodes must reach the gate as well.
        # None (default): the reference's rule alone.
        self._population_gate = None if population_gate is None else float(population_gate)
        # env_kw: DqlConfig fields of the simulated world beyond the Trainer's own arguments, e.g. config.AS_LAUNCHED (the parameters the
        # reference's manager node resolved under roslaunch: platform 1 m/s, observation noise 0.25 m / 0.1 m/s — tests/test_g14_gazebo.py)
        self._env_kw = dict(env_kw or {})
        # restart_after (episodes per env; None = never, the reference's behaviour): a level k >= 1 that has not been promoted after this many episodes
        # per env since its (re)start is started over — its slice of the tables is transferred from level k - 1 again (Eq. 31) and its visit counters
        # cleared, the deque and the population window emptied.  Why: with zero exploration above level 0 (pkg/trainer.py:112-114) and learning rates
        # at their floor, a level whose first few thousand greedy episodes settled on a slightly worse limit cycle STAYS there: 4 of 12 seeds sat at
        # 0.89-0.93 population success for their whole 768-episodes-per-env budget in round 4 while the other levels promoted within 4-80 episodes per
        # env (profiles/r4_bench_default_detail.json).  A restart is a fresh draw of that early phase; the promotion rule itself is untouched.
        self._restart_after = None if restart_after is None else float(restart_after)
        # step_back_after (restarts; None = never): a level k >= 2 still not promoted after this many restarts is not restarted again — the trainer steps BACK one level:
        # level k - 1 is learnt again from level k - 2 (transfer, cleared counters, the promotion rule as for any level) and level k after it.  Restarts redraw a level's
        # fixed point from the SAME table of the level below; when every draw fails it is that table the level cannot be learnt from (48 seeds of the bench recipe: a level
        # that is promoted at all needs at most 3 restarts, a level that fails fails all 8 attempts of its budget, profiles/r5_curriculum_48_seeds.jsonl).
        # The level learnt again gets step_back_after + 1 restart windows, never steps back itself (no cascade), and when it is not promoted in them it falls back to the
        # tables it WAS promoted with (the level above then goes on from those).  At most `max_step_backs` per run; the best attempt of a level survives its step-backs and
        # is what a finally exhausted budget hands over.
        self._step_back_after = None if step_back_after is None else int(step_back_after)
        self._max_step_backs = int(max_step_backs)
        if self._step_back_after is not None and (self._step_back_after < 1 or self._restart_after is None):
            raise ValueError("step_back_after needs restart_after and must be >= 1 restart")
        # (exploration above level 0 — eps 0.02 / 0.05 / 0.1 for a level's first 32 / 64 episodes per env — was tried and ends learning: no level above 0
        # promoted in 4 of 4 seeds, profiles/r5_curriculum_upper_level_exploration.jsonl; the reference's "no exploration above level 0" stands)
        # transfer_counts f (default 0 = the reference: a new level's visit counters start at 0, so its first visits learn with alpha = 1, 0.70, 0.57 ...):
        # level k starts with count[k] = f * count[k - 1] — the transferred values are refined with the learning rates they were learnt with
        self._transfer_counts = float(transfer_counts)
        # (also starting level k - 1's slice over at every / every other restart of level k — "deep restarts" — moves the coin, it does not load it: of the two
        # seeds whose level 4 never reaches the gate one is cured and another seed breaks at level 2, profiles/r5_curriculum_deep_restart.jsonl; not kept)
        if not 1 <= self._periods_per_launch <= 32 or self._chunk_steps % self._periods_per_launch:
            raise ValueError("periods_per_launch must be in 1..32 and divide chunk_steps")
        # device None: GPU LOCAL_RANK of a multi-rank launch (one process per GPU), GPU 0 of a single process; an explicit device wins
        self._comm = comm if comm is not None else RcclComm.from_env(device)  # None: single process
        self._device_arg = device
        self._device = int(device) if device is not None else (getattr(self._comm, "device", 0) if self._comm else 0)
        self._save_generation = 0  # checkpoints written by this Trainer (every rank counts the same ones)
        self._reducer_factory = reducer_factory
        self._rank = self._comm.rank if self._comm else 0
        self._world = self._comm.world if self._comm else 1
        self.history = []  # one record per finished curriculum level
        self._engine: Optional[Engine] = None
        self._progress: Optional[Dict[str, Any]] = None  # bookkeeping of the level in flight (what a checkpoint has to carry)
        self._resume: Optional[Dict[str, Any]] = None    # set by load(): progress to continue from
        self._alpha_cum = None

    # ---- schedules: pkg/trainer.py:88-138 ----
    def alpha(self, current_state_action: StateAction):
        counter = self._double_q_learning_agent.state_action_counter[current_state_action]
        if counter == 0:
            self._alpha = self._alpha_min
        else:
            # = np.max([np.float_power(1 / counter, omega), alpha_min]) of the reference (pkg/trainer.py:95) without building the list's array:
            # max() keeps a NaN first argument, as np.max does
            self._alpha = float(max(np.float_power(1 / (counter), self._omega), self._alpha_min))
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
        build = {k: getattr(self, "_" + k) for k in _BUILD_KEYS if k not in ("judge_envs", "device")}
        build["judge_envs"] = self._judge_envs_arg
        build["device"] = self._device_arg  # None stays None: a resumed multi-rank job again takes LOCAL_RANK
        return {"curriculum_steps": self._curriculum_steps, "working_curriculum_step": self._working_curriculum_step,
                "current_episode": self._current_episode, "curriculum_episode_count": self._curriculum_episode_count,
                "seed": self._seed, "alpha_min": self._alpha_min, "omega": self._omega, "gamma": self._gamma,
                "scale_modification_value": self._scale_modification_value, "t_max": self._t_max, "z_init": self._z_init,
                "f_ag": self._f_ag, "p_max": self._p_max, "success_rate": self._success_rate,
                "successive_successful_episodes": self._successive_successful_episodes, "max_num_episodes": self._max_num_episodes,
                "build": build, "world": self._world, "history": self.history, "progress": self._progress}

    def _env_state_file(self, rank: int) -> Path:
        return self._save_path / f"env_state_rank{rank}.npz"

    def _checkpoint_tag(self):
        """What ties the files of ONE checkpoint together: every rank is at the same (level, agent periods, chunk) when it saves, and
        counts the same checkpoints.  Stored in every env_state_rank*.npz and in trainer.json's progress."""
        pr = self._progress or {}
        return {"level": int(pr.get("level", -1)), "steps": int(pr.get("steps", -1)), "chunk_i": int(pr.get("chunk_i", -1)),
                "world": int(self._world), "generation": int(self._save_generation)}

    def save(self) -> None:
        """One checkpoint = env_state_rank<r>.npz of every rank + the three .npy tables + trainer.json, every file written as temp +
        os.replace, trainer.json LAST (and, with several ranks, after a barrier): a run killed anywhere in between leaves the previous
        trainer.json, whose tag the newer env-state files do not match — `load()` then resumes the level from its tables without them."""
        eng = self._engine
        if eng is not None and hasattr(eng, "publish_tables"):
            eng.publish_tables()  # checkpoint = table barrier: pending updates folded, acting tables = master tables (a resumed run starts so)
        self._pull_tables()
        self._save_generation += 1
        tag = self._checkpoint_tag()
        if eng is not None and self._checkpoint_env_state and self._progress is not None and hasattr(eng, "get_fields"):
            # every rank writes its own shard: simulator state of every env + the period index the RNG and tick schedule hang on
            self._save_path.mkdir(parents=True, exist_ok=True)
            reals, ints = eng.get_fields()
            f = self._env_state_file(self._rank)
            tmp = f.with_name(f".{f.name}.{os.getpid()}.tmp.npz")
            # dtype + state_layout: the two PIDs' filter fields mean different things per dtype (state_layout.py); a shard is only flown by a
            # context it was written for, or after the one mapping that exists (float64 histories -> float32 transposed states)
            np.savez(tmp, reals=reals.astype(np.float32 if self._dtype == F32 else np.float64), ints=ints, step_index=np.int64(eng.step_index()),
                     dtype=np.int64(self._dtype), state_layout=np.int64(STATE_LAYOUT_VERSION), **{f"tag_{k}": np.int64(v) for k, v in tag.items()})
            os.replace(tmp, f)
        if self._comm is not None and self._world > 1:
            self._comm.barrier()  # every rank's shard of this checkpoint is on disk before rank 0 publishes it
        if self._rank != 0:  # table replicas are identical after a sync: rank 0 writes
            return
        self._save_path.mkdir(parents=True, exist_ok=True)
        # the three .npy files keep the reference's layout, so the checkpoint they belong to is named NEXT to them, and the name brackets the
        # writes: "in_progress" before the first table, the plain tag after the last.  A run killed between two of the three table files leaves
        # the in-progress tag (load() warns: the tables may mix two checkpoints); killed between the tables and trainer.json it leaves newer
        # tables under the older progress counters, and load() says so
        def write_tag(t):
            tmp = self._save_path / f".tables.tag.json.{os.getpid()}.tmp"
            tmp.write_text(json.dumps(t))
            os.replace(tmp, self._save_path / "tables.tag.json")
        write_tag(dict(tag, in_progress=1))
        self._double_q_learning_agent.save(self._save_path)
        self._double_q_learning_agent.save(self._save_path / "..")
        write_tag(tag)
        st = self._state_dict()
        if st["progress"] is not None:
            st["progress"] = dict(st["progress"], tag=tag)
        tmp = self._save_path / f".trainer.json.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            json.dump(st, fh, indent=1)
        os.replace(tmp, self._save_path / "trainer.json")

    @staticmethod
    def load(assets_path: Path = ASSETS_PATH, **kw) -> "Trainer":
        """Latest run directory (named `dd-mm-YYYY HH:MM:SS`, pkg/utils.py) under `assets_path`.  Keywords override what the
        checkpoint holds (e.g. device=...)."""
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
        try:
            tables_tag = json.loads((run / "tables.tag.json").read_text())
        except (OSError, ValueError):
            tables_tag = None
        want_tag = (st.get("progress") or {}).get("tag")
        if tables_tag is not None and tables_tag.pop("in_progress", 0):
            warnings.warn(f"the tables in {run.name} were being written (checkpoint {tables_tag}) when the run stopped: the three .npy files may belong to two "
                          "different checkpoints — resuming with them as they are", RuntimeWarning)
        elif tables_tag is not None and want_tag is not None and tables_tag != want_tag:
            warnings.warn(f"the tables in {run.name} belong to checkpoint {tables_tag}, trainer.json to {want_tag}: the run was stopped between the two "
                          "writes — resuming with the newer tables under the older progress and episode counters", RuntimeWarning)
        build = dict(st.get("build", {"n_envs": st.get("n_envs", 4096), "mode": st.get("mode", "reference")}))
        build.update(kw)
        tr = Trainer(curriculum_steps=st["curriculum_steps"], double_q_learning_agent=agent, initial_curriculum_step=st["working_curriculum_step"],
                     seed=st["seed"], save_path=run, alpha_min=st["alpha_min"], omega=st["omega"], gamma=st["gamma"],
                     scale_modification_value=st["scale_modification_value"], t_max=st["t_max"], z_init=st["z_init"], f_ag=st["f_ag"],
                     p_max=st["p_max"], success_rate=st["success_rate"], successive_successful_episodes=st["successive_successful_episodes"],
                     max_num_episodes=st["max_num_episodes"], **build)
        tr._current_episode, tr._curriculum_episode_count, tr.history = st["current_episode"], st["curriculum_episode_count"], st["history"]
        tr._resume = st.get("progress")
        if tr._resume is not None and st.get("world", 1) != tr._world:
            raise ValueError(f"checkpoint was written by {st.get('world', 1)} rank(s), this job has {tr._world}: env shards would not line up")
        return tr

    # ---- device plumbing ----
    def _config(self, level: int) -> DqlConfig:
        return DqlConfig(working_curriculum_step=level, dtype=self._dtype, quirks=self._quirks,
                         t_max=self._t_max, z_init=self._z_init, f_ag=self._f_ag, p_max=self._p_max, init_sigma=self._p_max / 3,
                         gamma=self._gamma, alpha_min=self._alpha_min, alpha_omega=self._omega, fold_per_step=self._fold_per_step, **self._env_kw)

    def _push_tables(self):
        a = self._double_q_learning_agent
        qa, qb, cnt = a._padded()
        self._engine.set_tables(qa, qb, cnt)

    def _pull_tables(self):
        if self._engine is None:
            return
        qa, qb, cnt = self._engine.get_tables()
        self._double_q_learning_agent._unpad(np.asarray(qa).reshape(-1), np.asarray(qb).reshape(-1), np.asarray(cnt).reshape(-1))

    # ---- pkg/trainer.py:169-245 ----
    def _make_engine(self, cfg):
        sync = self._sync_period if self._sync_period is not None else (2 if (self._world > 1 or self._reducer_factory is not None) else None)
        if self._world > 1 or self._reducer_factory is not None or sync is not None:
            if self._chunk_steps % sync or sync % self._periods_per_launch:
                raise ValueError("chunk_steps must be a multiple of sync_period (checkpoints and promotions happen on synchronised tables) and sync_period of periods_per_launch")
            lo, hi = shard_range(self._n_envs, self._rank, self._world)
            eng = Engine(cfg, hi - lo, seed=self._seed, device=self._device, env_id_offset=lo)
            if self._periods_per_launch != 1:
                eng.set_option("periods_per_launch", self._periods_per_launch)
            if self._reducer_factory is not None:
                make = self._reducer_factory
            elif self._world > 1:
                make = self._comm.reducer
            else:
                make = LocalWindowReducer  # one rank on the windowed schedule: what N ranks with this sync period do
            return eng, ShardedRunner(eng, make(eng), sync)
        eng = Engine(cfg, self._n_envs, seed=self._seed, device=self._device)
        if self._periods_per_launch != 1:
            eng.set_option("periods_per_launch", self._periods_per_launch)
        return eng, ShardedRunner(eng, None)

    def _seed_level_counts(self, eng, k):
        _, _, cnt = eng.get_tables()
        cnt = np.asarray(cnt, dtype=np.float64).reshape(-1).copy()
        per_level = cnt.size // 5
        cnt[k * per_level:(k + 1) * per_level] = np.floor(self._transfer_counts * cnt[(k - 1) * per_level:k * per_level])
        eng.set_tables(count=cnt)

    def _chunk_counters(self, s, s_prev):
        by, by0 = s["by_code"], s_prev["by_code"]
        v = np.array([s["episodes"] - s_prev["episodes"], by["TERMINAL_SUCCESS"] - by0["TERMINAL_SUCCESS"],  # "Goal state reached" only (B17)
                      s["decisions"] - s_prev["decisions"], s["reward_sum"] - s_prev["reward_sum"]]
                     + [by.get(c, 0) - by0.get(c, 0) for c in _TERMINAL], dtype=np.float64)
        return self._comm.all_reduce_sum(v) if self._comm else v

    def _judge_layout(self):
        """Which bit columns of the (gathered) episode log are judged envs: global ids 0 .. judge_envs-1, rank by rank."""
        cnt = []
        for r in range(self._world):
            lo, hi = shard_range(self._n_envs, r, self._world)
            cnt.append(int(np.clip(self._judge_envs - lo, 0, hi - lo)))
        w = max(1, max((c + 63) // 64 for c in cnt))
        valid = np.concatenate([np.arange(w * 64) < c for c in cnt])
        return cnt[self._rank], w, valid

    def _judge_masks(self, done, goal, cnt, w):
        """this rank's log restricted to its judged envs, padded to w words per period"""
        P = done.shape[0]
        out = np.zeros((2, P, w), dtype=np.uint64)
        k = min(w, done.shape[1])
        keep = np.packbits(np.arange(k * 64) < cnt, bitorder="little").view(np.uint64)
        out[0, :, :k] = done[:, :k] & keep; out[1, :, :k] = goal[:, :k] & keep
        if self._comm:
            return self._comm.all_gather_masks(out[0], out[1])
        return out[0], out[1]

    def _mean_alpha(self, cnt_before, cnt_after):
        """Visit-weighted mean learning rate of the visits between two snapshots of state_action_counter: visit number c of a cell
        (0-based) is made at alpha(c) (pkg/trainer.py:88-110, B5)."""
        if self._alpha_cum is None:
            tab = self._config(0).alpha_table()
            self._alpha_cum = np.concatenate([[0.0], np.cumsum(tab)])
        cum, n_tab = self._alpha_cum, len(self._alpha_cum) - 1
        def total(c):  # sum of alpha(0 .. c-1)
            c = np.asarray(c, dtype=np.float64)
            inside = np.minimum(c, n_tab).astype(np.int64)
            return cum[inside] + (c - inside) * self._alpha_min
        visits = float(np.sum(cnt_after) - np.sum(cnt_before))
        if visits <= 0:
            return self._alpha
        return float((np.sum(total(cnt_after)) - np.sum(total(cnt_before))) / visits)

    def _load_env_state(self, eng, progress):
        """This rank's env-state file if it is of the checkpoint `progress` belongs to, else None: there is no file, a rank was killed
        between its shard and rank 0's trainer.json, the file is a leftover of an earlier level, the world size changed."""
        f = self._env_state_file(self._rank)
        if not (self._checkpoint_env_state and f.exists() and hasattr(eng, "set_fields")):
            return None
        z = np.load(f, allow_pickle=False)
        want = progress.get("tag")
        have = {k[4:]: int(z[k]) for k in z.files if k.startswith("tag_")}
        if want is None or have != {k: int(v) for k, v in want.items()}:
            warnings.warn(f"{f.name} belongs to checkpoint {have or 'without a tag'}, trainer.json to {want}: resuming level {progress.get('level')} "
                          "from its tables without the saved simulator state", RuntimeWarning)
            return None
        # which dtype's layout the shard holds (files written before round 5 carry no field: the array's own dtype, layout 1 = histories everywhere)
        src_dtype = int(z["dtype"]) if "dtype" in z.files else (F32 if z["reals"].dtype == np.float32 else 1)
        src_layout = int(z["state_layout"]) if "state_layout" in z.files else 1
        reals = convert_env_state(z["reals"].astype(np.float64), eng.field_names(), src_dtype, self._dtype, src_layout, bw_c=self._env_kw.get("bw_c", 1.0))
        if reals is None:
            warnings.warn(f"{f.name} holds the simulator state of a {'float32' if src_dtype == F32 else 'float64'} context (layout {src_layout}) and this run is "
                          f"{'float32' if self._dtype == F32 else 'float64'}: the float32 filter states do not determine the float64 histories — resuming level "
                          f"{progress.get('level')} from its tables without the saved simulator state", RuntimeWarning)
            return None
        z = {"reals": reals, "ints": z["ints"], "step_index": z["step_index"]}
        if z["ints"].shape[1] != eng.n:  # raised by _restore_env_state on EVERY rank, after the vote (a lone raise here would leave the others in the collective)
            self._env_state_error = f"{f} holds {z['ints'].shape[1]} envs, this rank's shard has {eng.n}"
            return None
        return z

    def _restore_env_state(self, eng, progress):
        """Simulator state of the envs from the checkpoint `progress` belongs to — on every rank, or on none (then the level's envs
        start over, its bookkeeping does not)."""
        self._env_state_error = None
        z = self._load_env_state(eng, progress)
        ok, bad = z is not None, self._env_state_error is not None
        if self._comm is not None and self._world > 1:
            votes = self._comm.all_reduce_sum([0.0 if ok else 1.0, 1.0 if bad else 0.0])
            ok, bad_anywhere = bool(votes[0] == 0.0), bool(votes[1] > 0.0)
        else:
            bad_anywhere = bad
        if bad_anywhere:  # a shard of another env count is a configuration error, not a missing file: every rank stops, together
            raise ValueError(self._env_state_error or "another rank's env-state shard holds a different env count than its engine (see that rank's message)")
        if not ok:
            return False
        eng.set_fields(np.asarray(z["reals"], dtype=np.float64), z["ints"])
        eng.set_step_index(int(z["step_index"]))
        return True

    def curriculum_training(self):
        if self._working_curriculum_step >= self._curriculum_steps:  # Trainer.load() of a finished run: nothing left to train
            return self.history
        t_start = time.perf_counter()
        resume, self._resume = self._resume, None
        cfg = self._config(self._working_curriculum_step)
        self._engine, runner = self._make_engine(cfg)
        self._push_tables()
        eng = self._engine
        ordered = self._promotion_rule == "ordered"
        if ordered:
            eng.episode_log_enable(self._chunk_steps)
        pw = PromotionWindow(self._successive_successful_episodes, self._success_rate)
        j_cnt, j_w, j_valid = self._judge_layout()
        # only the judged envs' words of the log cross the bus when the engine can restrict the read (16 B per env and chunk otherwise)
        import inspect
        read_log = getattr(eng, "episode_log_read", None)  # (only the ordered rule reads the log)
        if read_log is not None and "words" in inspect.signature(read_log).parameters:
            read_log = lambda: eng.episode_log_read(words=(j_cnt + 63) // 64)
        order = EpisodeOrder(j_valid.size, j_valid)
        first_level = self._working_curriculum_step
        k_next = first_level
        first_promotion_wall: Dict[int, float] = {}
        step_backs, best_of_level = 0, {}  # (step_back_after) steps back taken so far; best attempt of a level over its lineages
        promoted_snap: Dict[int, Any] = {}  # level -> (tables, history entry) when the rule promoted it: what a failed re-learning of the level falls back to
        relearning = None                   # the level being learnt again after a step back (it never steps back itself, and gets a short budget)
        while k_next < self._curriculum_steps:
            self._working_curriculum_step = k = k_next
            k_next = k + 1
            resumed = resume is not None and k == first_level and resume.get("level") == k
            resume = resume if resumed else None  # (a level entered again after a step back starts afresh)
            if resumed:
                # continue the interrupted level: no transfer (it was applied when the level started), tables as checkpointed
                have_envs = self._restore_env_state(eng, resume)
                if not have_envs:
                    eng.set_curriculum(k)  # no simulator state saved: the level's envs start over, its bookkeeping does not
                pr = resume
                window = deque(tuple(w) for w in pr["window"])
                pw.reset(); order.reset()
                if have_envs:
                    pw.set_state(pr["pw"]); order.set_state(pr["order"])
                episodes, steps, chunk_i = int(pr["episodes"]), int(pr["steps"]), int(pr["chunk_i"])
            else:
                if self._mode == "paper" and k >= 1:
                    eng.transfer(k, self.transfer_learning_ratio(k))  # Eq. 31 as intended: the new level starts from the previous one
                    if self._transfer_counts > 0.0:
                        self._seed_level_counts(eng, k)
                eng.set_curriculum(k)  # "Create a new environment to update limits" (:175-183)
                window = deque()  # aggregate rule: (episodes, goal-state successes) per chunk, trimmed to the most recent >= 100 episodes
                pw.reset(); order.reset()
                episodes, steps, chunk_i = 0, 0, 0
            s_prev = eng.stats()
            counts = eng.get_counts if hasattr(eng, "get_counts") else (lambda: eng.get_tables()[2])
            cnt_prev = np.asarray(counts(), dtype=np.float64).copy()
            t_level = time.perf_counter()
            promoted = False
            promoted_at = None
            restarts, restart_base = (int(pr.get("restarts", 0)), int(pr.get("restart_base", 0))) if resumed else (0, 0)
            best_attempt = best_of_level.get(k)  # (population success rate, tables) of the best attempt given up so far (not checkpointed: a resumed run starts collecting again)
            stalled = False
            info: Dict[str, Any] = {}
            # a level learnt AGAIN after a step back gets (step_back_after + 1) restart windows, not the whole budget: it has been promoted from these tables before
            level_budget = self._max_num_episodes
            if relearning == k and not resumed:
                level_budget = min(level_budget, int((self._step_back_after + 1) * self._restart_after * self._n_envs))
            while episodes < level_budget:
                sched_episode = int(episodes / self._eps_episode_scale)
                eps = max(self.exploration_rate(sched_episode, k), self._eps_floor)  # scale 1, floor 0: the reference schedule
                if self._eps_tail is not None and k == 0 and sched_episode >= 2000 and episodes >= self._eps_tail_after * self._n_envs:
                    eps = self._eps_tail  # what follows the schedule's decay (the reference stays at 0.01 for the rest of level 0)
                runner.train_steps(self._chunk_steps, eps)
                steps += self._chunk_steps
                s = eng.stats()
                cc = self._chunk_counters(s, s_prev)
                new_eps, new_ok, new_dec, new_rew = int(cc[0]), int(cc[1]), cc[2], cc[3]
                s_prev = s
                hit = None
                if ordered:
                    done, goal = self._judge_masks(*read_log(), j_cnt, j_w)
                    hit = pw.push_flags(order.push(done, goal))
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
                cnt_now = np.asarray(counts(), dtype=np.float64)
                self._alpha = self._mean_alpha(cnt_prev, cnt_now)
                cnt_prev = cnt_now
                info = {"Curent episode": episodes, "Remaining episodes": self._max_num_episodes - episodes + 1, "Exploration rate": eps,
                        "Learning rate": self._alpha, "Success rate": rate, "Mean reward": new_rew / max(1, new_dec),
                        "Cumulative reward": new_rew / max(1, new_eps), "Agent periods": steps, "Curriculum step": k,
                        "Termination condition": {c: int(v) for c, v in zip(_TERMINAL, cc[4:])}}
                chunk_i += 1
                done_level = (hit is not None) if ordered else (rate > self._success_rate)
                if done_level and self._population_gate is not None and rate < self._population_gate:
                    done_level = False  # the judged envs' window passed, the population has not arrived: the level goes on (the deque keeps sliding)
                if (not done_level and self._restart_after is not None and k >= 1 and self._mode == "paper"
                        and episodes - restart_base >= self._restart_after * self._n_envs and episodes < level_budget):
                    # start the level over (see __init__): every rank does the same, at a chunk boundary = on synchronised tables
                    qa_, qb_, cnt = (np.asarray(t, dtype=np.float64).reshape(-1).copy() for t in eng.get_tables())
                    if best_attempt is None or rate > best_attempt[0]:  # the attempt given up may still be the best one the budget buys
                        best_attempt = (rate, qa_, qb_, cnt.copy())
                    if (self._step_back_after is not None and k >= 2 and k - 1 >= first_level and restarts >= self._step_back_after and step_backs < self._max_step_backs
                            and relearning is None):
                        stalled = True  # every redraw from this table of level k - 1 failed: step back instead of restarting again (see __init__)
                        break
                    per_level = cnt.size // 5
                    cnt[k * per_level:(k + 1) * per_level] = np.floor(self._transfer_counts * cnt[(k - 1) * per_level:k * per_level]) if self._transfer_counts > 0.0 else 0.0
                    eng.set_tables(count=cnt)
                    eng.transfer(k, self.transfer_learning_ratio(k))
                    window.clear(); pw.reset(); order.reset()
                    if ordered:
                        read_log()  # episodes already logged belong to the attempt that was given up
                    cnt_prev = cnt
                    restarts += 1; restart_base = episodes
                if chunk_i % self._checkpoint_every == 0 and not done_level:
                    self._progress = {"level": k, "episodes": episodes, "steps": steps, "chunk_i": chunk_i, "window": [list(w) for w in window],
                                      "pw": pw.get_state(), "order": order.get_state(), "restarts": restarts, "restart_base": restart_base}
                    self.save()
                self.log(info)
                if done_level:
                    self._successes = deque([], maxlen=self._successive_successful_episodes)
                    promoted = True
                    if hit is not None:  # which judged episode (generation order) filled the reference's deque to > success_rate
                        promoted_at = {"judged_episode": hit + 1}
                    break
                if self._max_steps_per_level is not None and steps >= self._max_steps_per_level:
                    break
            if stalled:
                best_of_level[k] = best_attempt
                step_backs += 1
                cnt = np.asarray(eng.get_tables()[2], dtype=np.float64).reshape(-1).copy()
                per_level = cnt.size // 5
                cnt[(k - 1) * per_level:(k + 1) * per_level] = 0.0  # levels k - 1 and k learn again from fresh counters; their slices are rewritten by the transfers
                eng.set_tables(count=cnt)
                while self.history and self.history[-1]["level"] >= k - 1:  # one entry per level: level k - 1's is written again when it ends again
                    self.history.pop()
                self._progress = None
                relearning = k - 1
                k_next = k - 1
                continue
            exhausted = not promoted and episodes >= level_budget
            if exhausted and best_attempt is not None and best_attempt[0] > (info.get("Success rate") or 0.0):
                # the budget ran out in a later, worse attempt: the level hands over the tables of its best one (same rule as the reference's budget
                # hand-over, pkg/trainer.py:187 — it just does not hand over a restart that had barely begun)
                eng.set_tables(best_attempt[1], best_attempt[2], best_attempt[3])
                info["Success rate"] = best_attempt[0]
            if promoted and k not in first_promotion_wall:
                first_promotion_wall[k] = time.perf_counter() - t_start
            fell_back = False
            if relearning == k:
                relearning = None
                if not promoted and k in promoted_snap:
                    # learnt again and not promoted within its short budget: the level keeps the tables (and the history entry) it WAS promoted with; the level
                    # above goes on from those with plain restarts
                    eng.set_tables(*promoted_snap[k][0])
                    self.history.append(dict(promoted_snap[k][1], step_backs=step_backs, wall_since_start_s=time.perf_counter() - t_start))
                    promoted, exhausted, fell_back = True, False, True
            if not fell_back:
                entry = {"level": k, "promoted": promoted, "exhausted": exhausted, "promoted_at": promoted_at, "restarts": restarts, "step_backs": step_backs, "episodes": episodes,
                         "agent_periods": steps, "success_rate": info.get("Success rate"),
                         "wall_s": time.perf_counter() - t_level, "wall_since_start_s": time.perf_counter() - t_start,
                         # (with step backs a level can end more than once: when the rule FIRST promoted it — "stage k + 1 entered")
                         "wall_first_promoted_s": first_promotion_wall.get(k)}
                self.history.append(entry)
                if promoted and self._step_back_after is not None:
                    promoted_snap[k] = (tuple(np.asarray(t, dtype=np.float64).reshape(-1).copy() for t in eng.get_tables()), dict(entry))
            if self._mode == "reference":
                # transfer AFTER finishing level k: Q[k] = Q[k-1] * ratio, k = 0 wraps (B6, pkg/trainer.py:237-243)
                eng.transfer(k, self.transfer_learning_ratio(k))
            self._progress = None  # between levels: a resumed run starts the next level from its beginning
            try:  # this level's simulator state is of no use to the next one
                self._env_state_file(self._rank).unlink()
            except OSError:
                pass
            if promoted or exhausted:
                self._working_curriculum_step = min(k + 1, self._curriculum_steps)  # what a checkpoint taken now resumes at
                self.save()
                self._working_curriculum_step = k
            else:  # max_steps_per_level (build-specific bound) hit: stop here
                self.save()
                break
        fh = getattr(self, "_log_fh", None)
        if fh is not None and not fh.closed:
            fh.close()
        return self.history

    # ---- pkg/trainer.py:247-303: scalar log with the reference's tag names (CSV instead of one TensorBoard file per episode) ----
    def log(self, info: Dict[str, Any], clean=False):
        if self._rank != 0:
            return
        fh = getattr(self, "_log_fh", None)
        if fh is None or fh.closed:  # opened once per Trainer (a row per chunk: open + close per row showed up in the loop's profile), flushed per row
            path = self._save_path / "logs"
            path.mkdir(parents=True, exist_ok=True)
            f = path / "scalars.csv"
            new = not f.exists()
            fh = self._log_fh = open(f, "a", newline="")
            if new:
                csv.writer(fh).writerow(LOG_COLUMNS)
        term = info.get("Termination condition") or {}
        csv.writer(fh).writerow([info.get("Curriculum step"), self._curriculum_episode_count, info.get("Curent episode"), info.get("Agent periods"),
                                 info.get("Success rate"), info.get("Cumulative reward"), info.get("Exploration rate"), info.get("Learning rate"),
                                 info.get("Mean reward")] + [term.get(c, 0) for c in _TERMINAL])
        fh.flush()
        if not self._quiet:
            print(" | ".join(f"{k}: {v}" for k, v in info.items()), flush=True)
