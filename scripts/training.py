#!/usr/bin/env python3
"""Counterpart of the reference's scripts/training.py (rospy.init_node; Trainer().curriculum_training()):
curriculum training of the tabular Double-Q landing agent on one MI355X.

    python scripts/training.py [--envs 4096] [--mode reference|paper] [--out DIR] [--max-steps-per-level N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        scripts/training.py --envs 262144 --recipe bench         # BASELINE configs[3]: one rank per GPU, RCCL table exchange every 16 periods
    python scripts/training.py --envs 32768 --recipe bench       # its per-GPU share on one GPU: the curriculum leg of bench.py (stage 4 after ~1.0 s)
(torch.distributed.run is only the process launcher here; any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
MASTER_PORT will do.  The ranks themselves never import PyTorch.)
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--mode", default="reference", choices=["reference", "paper"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--max-steps-per-level", type=int, default=None)
    ap.add_argument("--max-episodes", type=int, default=None, help="episodes per level before the next level starts (reference: 50000 for one env; default max(50000, 64 per env))")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--fold-per-step", type=int, default=1, help="1 (default): one learning-rate step per launch towards the mean target; 0: m sequential visits")
    ap.add_argument("--eps-floor", type=float, default=0.0)
    ap.add_argument("--success-rate", type=float, default=0.96, help="promotion threshold (reference: 0.96, pkg/trainer.py:25)")
    ap.add_argument("--promotion-rule", default="ordered", choices=["ordered", "aggregate"])
    ap.add_argument("--sync-period", type=int, default=None, help="agent periods between table exchanges (default: none on one GPU, 2 on several; given on one GPU it runs the same windowed schedule)")
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--t-max", type=int, default=20)
    ap.add_argument("--judge-envs", type=int, default=1, help="whose episodes feed the promotion deque (1 = the reference's own situation, Trainer default)")
    ap.add_argument("--periods-per-launch", type=int, default=1, help="agent periods per kernel launch, 1..32 (must divide --chunk and --sync-period)")
    ap.add_argument("--quirks", type=lambda v: int(v, 0), default=None, help="override of the mode's quirk set (include/dql.h DQL_Q_*), e.g. 0x60")
    ap.add_argument("--eps-tail", type=float, default=None, help="exploration rate of level 0 once the reference's schedule has decayed (default: the reference's 0.01 floor)")
    ap.add_argument("--eps-tail-after", type=float, default=0.0, help="... from this many episodes per env on")
    ap.add_argument("--recipe", default=None, choices=["bench"], help="bench: the trainer settings of bench.py's curriculum leg (bench.CURRICULUM_KW incl. the population gate, mode paper, sync 16, 768 episodes per env and level)")
    ap.add_argument("--window", type=int, default=100, help="successive_successful_episodes (reference: 100)")
    ap.add_argument("--as-launched", action="store_true",
                    help="fly the world the reference's manager node RESOLVED under roslaunch (platform 1 m/s, observation noise 0.25 m / 0.1 m/s: config.AS_LAUNCHED, "
                         "golden G14) instead of the launch file's literal values (1.6 m/s, no noise)")
    ap.add_argument("--restart-after", type=float, default=None, help="a level above 0 not promoted after this many episodes per env is started over from the level below (default: never, the reference; --recipe bench: 96)")
    ap.add_argument("--step-back-after", type=int, default=None, help="a level above 1 still not promoted after this many restarts goes back one level instead of restarting again (needs --restart-after; default: never; --recipe bench: 3)")
    ap.add_argument("--population-gate", type=float, default=None, help="promotion also needs this success rate of ALL envs (default: the reference's deque alone; --recipe bench: 0.94)")
    ap.add_argument("--attempts", type=int, default=1, help="whole curricula to train (seed + 7919 j) until one has every level promoted by the rule and lands >= --accept-touchdown of 4 096 greedy "
                                                          "episodes (dql_multirotor_landing_amd/attempts.py); none accepted: the best seen is kept.  1 (default): one run, the reference's situation; bench.py: 6")
    ap.add_argument("--accept-touchdown", type=float, default=0.875, help="default: what the reference's own stage-4 tables land in such a batch")
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    import os
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    import __graft_entry__ as g
    g.build_hip()
    from dql_multirotor_landing_amd.config import F32, F64
    from dql_multirotor_landing_amd.trainer import Trainer
    # one rank per GPU: the Trainer picks RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* up by itself (comm.RcclComm.from_env, RCCL inside libdql_hip.so)
    extra = dict(periods_per_launch=a.periods_per_launch, quirks=a.quirks, eps_tail=a.eps_tail, eps_tail_after=a.eps_tail_after)
    if a.recipe == "bench":
        import bench
        extra = dict(bench.CURRICULUM_KW)
        a.mode, a.sync_period, a.judge_envs = "paper", bench.CURRICULUM_SYNC, extra.pop("judge_envs")
        a.max_episodes = a.max_episodes or bench.CURRICULUM_BUDGET_PER_ENV * a.envs
    if a.as_launched:
        from dql_multirotor_landing_amd.config import AS_LAUNCHED
        extra["env_kw"] = dict(AS_LAUNCHED)
    if a.population_gate is not None:
        extra["population_gate"] = a.population_gate
    if a.restart_after is not None:
        extra["restart_after"] = a.restart_after
    if a.step_back_after is not None:
        extra["step_back_after"] = a.step_back_after
    from dql_multirotor_landing_amd.attempts import SELECTION_SEED, attempt_seed, curriculum_attempts
    from dql_multirotor_landing_amd.comm import RcclComm
    comm = RcclComm.from_env(local) if world > 1 else None  # ONE communicator for every attempt

    def make_trainer(j):
        out = a.out if (a.attempts == 1 or a.out is None) else f"{a.out}/attempt{j}"
        return Trainer(n_envs=a.envs, mode=a.mode, save_path=out, dtype=F32 if a.dtype == "f32" else F64, chunk_steps=a.chunk, device=local if world > 1 else None,
                       promotion_rule=a.promotion_rule, sync_period=a.sync_period, curriculum_steps=a.levels, t_max=a.t_max, judge_envs=a.judge_envs,
                       successive_successful_episodes=a.window, seed=attempt_seed(a.seed, j), comm=comm,
                       max_steps_per_level=a.max_steps_per_level, max_num_episodes=a.max_episodes, quiet=not a.verbose, fold_per_step=a.fold_per_step, eps_floor=a.eps_floor, success_rate=a.success_rate, **extra)

    if a.attempts == 1:
        tr = make_trainer(0)
        hist = tr.curriculum_training()
        out = {"history": hist, "save_path": str(tr._save_path), "world": world}
    else:
        from dql_multirotor_landing_amd.config import Q_PAPER, Q_REFERENCE
        from dql_multirotor_landing_amd.evaluation import landing_score

        def score(t):
            q = Q_PAPER if a.mode == "paper" else Q_REFERENCE  # (the worlds scripts/simulation.py --mode flies)
            return landing_score(t._double_q_learning_agent._padded(), 4096, a.levels - 1, seed=SELECTION_SEED, device=t._device, quirks=q)
        res = curriculum_attempts(make_trainer, score, max_attempts=a.attempts, accept_touchdown=a.accept_touchdown, comm=comm, rank=rank)
        tr, hist = res["trainer"], res["history"]
        out = {"history": hist, "save_path": str(tr._save_path), "world": world, "chosen_attempt": res["chosen"], "accepted": res["accepted"], "attempts": res["attempts"]}
    if rank == 0:
        print(json.dumps(out, indent=1))
    if tr._comm is not None:
        tr._comm.barrier()
