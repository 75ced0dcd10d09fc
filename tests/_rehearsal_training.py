#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the sharded Trainer with a torch/gloo communicator (tests/_torch_comm.py) instead of RCCL, so that
two ranks can share ONE GPU (RCCL wants a GPU per rank).  Same arguments as scripts/training.py; launched by
torch.distributed.run from tests/test_gpu_dropin.py."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    for name, typ, dflt in (("--envs", int, 4096), ("--chunk", int, 64), ("--sync-period", int, 2), ("--max-episodes", int, None), ("--levels", int, 5),
                            ("--t-max", int, 20), ("--window", int, 100), ("--success-rate", float, 0.96), ("--judge-envs", int, 4096)):
        ap.add_argument(name, type=typ, default=dflt)
    ap.add_argument("--mode", default="paper")
    ap.add_argument("--out", default=None)
    ap.add_argument("--attempts", type=int, default=1)
    ap.add_argument("--accept-touchdown", type=float, default=0.875)
    a = ap.parse_args()
    import torch  # torch first: it brings its own HIP runtime, which has to be the one the process uses
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _torch_comm import TorchComm
    from dql_multirotor_landing_amd.trainer import Trainer
    comm = TorchComm()

    def make(j):
        from dql_multirotor_landing_amd.attempts import attempt_seed
        return Trainer(n_envs=a.envs, mode=a.mode, save_path=a.out if a.attempts == 1 else f"{a.out}/attempt{j}", chunk_steps=a.chunk, device=0, sync_period=a.sync_period,
                       curriculum_steps=a.levels, t_max=a.t_max, judge_envs=a.judge_envs, successive_successful_episodes=a.window, max_num_episodes=a.max_episodes,
                       success_rate=a.success_rate, comm=comm, seed=attempt_seed(42, j))
    extra = {}
    if a.attempts == 1:
        tr = make(0)
        hist = tr.curriculum_training()
    else:  # scripts/training.py --attempts: rank 0 flies the selection batch, every rank gets the decision
        from dql_multirotor_landing_amd.attempts import SELECTION_SEED, curriculum_attempts
        from dql_multirotor_landing_amd.evaluation import landing_score
        res = curriculum_attempts(make, lambda t: landing_score(t._double_q_learning_agent._padded(), 256, a.levels - 1, seed=SELECTION_SEED, device=0),
                                  max_attempts=a.attempts, accept_touchdown=a.accept_touchdown, comm=comm, rank=rank)
        tr, hist = res["trainer"], res["history"]
        extra = {"chosen_attempt": res["chosen"], "accepted": res["accepted"], "attempts": res["attempts"]}
        # every rank must have reached the same decision: the ranks' chosen indices summed = world x rank 0's
        assert comm.all_reduce_sum([float(res["chosen"])])[0] == world * res["chosen"]
    if rank == 0:
        print(json.dumps({"history": hist, "save_path": str(tr._save_path), "world": world, **extra}, indent=1))
    dist.barrier()
    dist.destroy_process_group()
