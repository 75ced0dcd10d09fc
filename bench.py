#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused UAV-landing + tabular Double-Q training step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4] [--envs E] [--dtype f32|f64] [--sync-period S]

One "step" = one agent period (1/22.92 s of simulated time: 21-22 physics ticks + MDP + TD update) for every env of every rank.

Workload (`--config`, default 4 at EVERY N — BASELINE.json quotes its whole-node metric on configs[4]; its per-GPU share fits one GPU):
  4  configs[4] share: 1 048 576 envs / 8 GPUs = 131 072 envs per GPU, per-env randomised sinusoidal platforms (r_x ~ U(1,3) m,
     t_x ~ U(0.8,1.6) m/s) + observation noise (0.25 m, 0.1 m/s, Kalman R = 0.1^2), x-axis MDP, curriculum step 0, float32 dynamics,
     int32 packed table index; algorithmic 328 B per env-step.  `--gpus 8` = 1 048 576 global envs.
  3  configs[3] share: 262 144 / 8 = 32 768 envs per GPU, x-axis, shared platform (320 B); its full 0 -> 4 curriculum is the
     curriculum leg of EVERY run (`--curriculum-envs`, default 32 768 per GPU), whatever the throughput config.
  2  configs[2]: 65 536 envs, joint x + y MDP (400 B), one GPU.
  1  configs[1]: 4 096 envs, x-axis (320 B), one GPU — 64 waves on a 1 024-SIMD chip: reported in every single-GPU run as the
     secondary `small_batch` block with its own roofline.
State is resident in HBM before the timed region.  N > 1: one process per GPU — started by torch.distributed.run (RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_* in the environment) or, without a launcher, by this script itself (`--gpus N` spawns N child ranks before
anything touches a GPU and fails if they cannot each have one); weak scaling, E envs per GPU, env shards with global env ids, int64
accumulator all-reduce every --sync-period steps by RCCL inside libdql_hip.so.  No PyTorch anywhere in this file.
`value` counts env-steps = (env, period) pairs in which an action was taken (reset periods are not counted).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_VALU_PEAK_TFLOPS = 157.3

# algorithmic bytes per env-step = persistent per-env state words x 4 B, read once + written once per agent step
ALGO_BYTES = {"x": 320, "x_per_env_platform": 328, "two_axis": 400}
PRESETS = {
    1: dict(envs=4096, two_axis=0, randomize_platform=0, noise=0, tag="configs[1]", short="4 096 envs, x-axis MDP, level 0, shared rpm platform", what="4 096 vectorised envs, x-axis MDP, curriculum step 0, shared rpm platform r = 2 m, omega = 0.8 rad/s"),
    2: dict(envs=65536, two_axis=1, randomize_platform=0, noise=0, tag="configs[2]", short="65 536 envs, joint x + y 2-axis MDP, level 0", what="65 536 envs, joint x + y 2-axis MDP, curriculum step 0"),
    3: dict(envs=32768, two_axis=0, randomize_platform=0, noise=0, tag="configs[3] share", short="32 768 envs per GPU (262 144 / 8), x-axis MDP, level 0, shared rpm platform", what="262 144 envs / 8 GPUs = 32 768 envs per GPU, x-axis MDP, shared rpm platform (the curriculum leg runs its full 0 -> 4 schedule)"),
    4: dict(envs=131072, two_axis=0, randomize_platform=1, noise=1, tag="configs[4] share", short="131 072 envs per GPU (1 048 576 / 8), per-env sinusoidal platforms + obs noise, x-axis MDP, level 0, f32 / int32 index",
            what="1 048 576 envs / 8 GPUs = 131 072 envs per GPU, per-env randomised sinusoidal platforms (r_x ~ U(1,3) m, t_x ~ U(0.8,1.6) m/s) + observation noise "
                 "0.25 m / 0.1 m/s with Kalman R = 0.1^2, x-axis MDP, curriculum step 0, fp32 dynamics / int32 packed table index"),
}


def algo_bytes(two_axis: int, randomize_platform: int) -> int:
    return ALGO_BYTES["two_axis"] if two_axis else (ALGO_BYTES["x_per_env_platform"] if randomize_platform else ALGO_BYTES["x"])


def lib_source_sha16() -> str:
    """Hash of the sources libdql_hip.so is built from: committed PMC figures (profiles/*_traffic.json, *_pmc_sq_summary.json) carry
    the hash of the kernel they measured, and a figure is only quoted for the library that produced it."""
    h = hashlib.sha256()
    for f in ("dql_multirotor_landing_amd/csrc/dql_hip.hip", "dql_multirotor_landing_amd/csrc/dql_device.hpp", "dql_multirotor_landing_amd/csrc/dql_refk.inc", "include/dql.h"):
        h.update((ROOT / f).read_bytes())
    return h.hexdigest()[:16]


def flavour_key(envs: int, P: int, two_axis: int, randomize_platform: int, noise: int) -> str:
    return f"{envs}_p{P}" + ("_2axis" if two_axis else ("_cfg4" if (randomize_platform and noise) else ""))


def committed_profile(kind: str):
    """the newest committed PMC summary of `kind` ("traffic" | "pmc_sq_summary") whose source stamp is this library's, else (None, why)"""
    sha = lib_source_sha16()
    for f in sorted((ROOT / "profiles").glob(f"r*_{kind}.json"), reverse=True):
        try:
            t = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        stamp = t.get("source_sha16")
        if stamp == sha:
            return t, f.name
    return None, f"no committed profiles/r*_{kind}.json carries this library's source hash {sha} (PMC passes not re-run since the kernel changed)"


def cpu_baseline(envs: int, steps: int, dtype: int, cfg_kw: dict):
    """The CPU oracle (a port of the same fused step: oracle/dql_oracle.c) on the host cores, bounded sample of the SAME workload
    flavour: all cores (OpenMP over envs) as the headline value, one thread next to it."""
    from dql_multirotor_landing_amd.config import DqlConfig
    from oracle.oracle import Oracle

    def run(threads, n_steps):
        o = Oracle(DqlConfig(dtype=dtype, **cfg_kw), envs, seed=42, n_threads=threads)
        o.train_steps(3, 1.0)
        d0 = o.stats_dict()["decisions"]
        t0 = time.perf_counter()
        o.train_steps(n_steps, 1.0)
        dt = time.perf_counter() - t0
        return (o.stats_dict()["decisions"] - d0), dt

    # usable host cores: affinity mask, capped by the cgroup CPU quota (a GPU box exposes every core of the host but grants a share)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 16))
    d1, t1 = run(1, steps)
    dn, tn = run(cores, steps * min(cores, 16))
    return {"value": dn / tn, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{envs} envs of the headline workload flavour x {steps * min(cores, 16)} agent periods ({dn} env-steps, {tn:.1f} s), OpenMP over envs on {cores} threads, same dtype, gcc -O2 -fopenmp",
            "single_thread_value": d1 / t1, "single_thread_sample": f"{envs} envs x {steps} agent periods ({d1} env-steps, {t1:.1f} s)"}


# Trainer keywords of the curriculum leg beyond the defaults (kept in one place: reported in bench_detail.json).  quirks 0x60 = paper-mode MDP
# (reward / observation quirks repaired, the reference's success counter kept) + the reference's own update rule (Q_table_a only, B1/B2);
# 16 agent periods per launch and table exchange every 16; eps_tail: level 0 follows the reference's exploration schedule, keeps its 0.01
# floor for the first 192 episodes per env and then stops exploring (profiles/r3_level0_eps_tail.jsonl).
# Promotion (round 4, profiles/r4_curriculum_gate_sweep.jsonl, 12 seeds per row): the reference's deque (100 episodes, > 0.96) is fed by the
# episodes of 64 judged envs in generation order and is NECESSARY but no longer sufficient — the success rate of ALL envs over the most recent
# chunks must reach population_gate = 0.94 as well (uniform starts cap the population near 0.95: a platform receding at 1.6 m/s from 4.5 m away
# cannot be caught; a gate of 0.95 promotes all five levels in 1 seed of 12).  Round 3's rule (2 judged envs, no gate) promoted at population
# rates down to 0.90; with the gate no promotion happens below 0.94, all five levels promote in 8 seeds of 12 (round 3: 6), the stage-4 tables
# reach goal-hold 0.943 / touchdown 0.878 on average and 0.929 / 0.829 in the worst seed (round 3: 0.940 / 0.872, worst 0.889 / 0.781).
# Round 5: restart_after = 96.  A level above 0 that has not been promoted after 96 episodes per env is started over (slice re-transferred from the level below,
# visit counters cleared, promotion windows emptied; the rule itself untouched).  Why: without exploration above level 0 (the reference's schedule; exploring there
# ends learning, profiles/r5_curriculum_upper_level_exploration.jsonl) a level SITS on whatever limit cycle its first greedy episodes found —
# profiles/r5_level_series_seed8.jsonl: 0.888 +- 0.004 population success for 700 episodes per env — and every restart lands on another one.  12 seeds,
# this kernel: all five levels by the rule in 6 seeds without restarts, in 10 with (profiles/r5_curriculum_restart_sweep.jsonl); the two that remain never see
# a level-4 attempt above 0.94 and hand over the best attempt they saw when the budget runs out.
CURRICULUM_KW: dict = {"quirks": 0x60, "judge_envs": 64, "periods_per_launch": 16, "eps_tail": 0.0, "eps_tail_after": 192, "population_gate": 0.94, "restart_after": 96, "step_back_after": 3}
CURRICULUM_BUDGET_PER_ENV = 768  # episodes per env and level before the next level starts anyway (the reference: 50 000 episodes of ONE env)
CURRICULUM_SYNC = 16
# Whole curricula per seed (dql_multirotor_landing_amd/attempts.py): a run's landing quality is a draw decided below level 4 — 48 single runs of this recipe end between 0.79 and
# 0.94 touchdown, 35 of them with all five levels by the rule and >= 0.85 (profiles/r5_curriculum_48_seeds.jsonl) — and a run costs two seconds, so a seed trains again (seed + 7919 j)
# until an attempt has all five levels by the rule and lands >= 0.875 of 4 096 greedy episodes of a selection batch (what the reference's own stage-4 tables land); the line reports the first attempts beside the chosen ones.
CURRICULUM_ATTEMPTS = 6
CURRICULUM_ACCEPT_TOUCHDOWN = 0.875  # the touchdown rate of the reference's own stage-4 tables in the same kind of batch (0.876, `reference_assets` in the line)
# tabular RL is seed-noisy (per seed: goal-hold 0.87-0.96, touchdown 0.70-0.95, profiles/r3_curriculum_p8_p16_judge_sweep.jsonl): twelve
# full curricula, each reported; 2 / 4 judged envs and 8 / 16 periods per launch are all within that noise of each other
CURRICULUM_SEEDS = (42, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11)


def curriculum_leg(args, comm, world, rank, dev_index, dtype):
    """Second half of BASELINE.json's metric: wall-clock to curriculum stage 4, on BASELINE configs[3]'s size (32 768 envs per GPU by
    default, `--curriculum-envs`).  The Trainer's loop (reference promotion rule: 100-episode deque > 0.96, or the level's episode budget
    runs out) sharded over the ranks of this job, then greedy roll-outs of the resulting stage-4 tables next to the reference's own (rank 0).
    Same table schedule (sync period and periods per launch) at any N, one GPU included, so the figures of a scaling sweep are the
    same run on more hardware.  Never fails the bench."""
    import tempfile
    try:
        from dql_multirotor_landing_amd.config import Q_PAPER
        from dql_multirotor_landing_amd.attempts import SELECTION_SEED, attempt_seed, curriculum_attempts
        from dql_multirotor_landing_amd.evaluation import landing_score
        from dql_multirotor_landing_amd.trainer import Trainer
        sys.path.insert(0, str(ROOT / "scripts"))
        import simulation

        def greedy(tables):
            h = simulation.evaluate(tables, 4096, 4, flavour="simulation", quirks=Q_PAPER, device=dev_index)
            g_ = simulation.evaluate(tables, 4096, 4, flavour="training", quirks=Q_PAPER, device=dev_index)
            return {"touchdown_rate": h["TERMINAL_CONTACT"] / 4096, "goal_hold_rate": g_["TERMINAL_SUCCESS"] / 4096}

        n_global = args.curriculum_envs * world
        # the reference's 50 000-episode budget per level assumes ONE env; with N envs at once it has to cover a few
        # generations of all of them, or a level ends before most envs have finished an episode (Trainer default)
        budget = max(args.curriculum_budget, CURRICULUM_BUDGET_PER_ENV * n_global)
        runs = []
        for seed in CURRICULUM_SEEDS[:args.curriculum_seeds]:
            with tempfile.TemporaryDirectory() as d:
                def make_trainer(j):
                    return Trainer(mode="paper", n_envs=n_global, device=dev_index, dtype=dtype, save_path=Path(d) / f"run{j}", chunk_steps=64, sync_period=CURRICULUM_SYNC,
                                   max_num_episodes=budget, checkpoint_every=10**9, comm=comm, seed=attempt_seed(seed, j), **CURRICULUM_KW)

                def score(tr):  # (rank 0) greedy landings of the attempt's tables on the SELECTION batch; the figures reported below fly seed 123
                    return landing_score(tr._double_q_learning_agent._padded(), 4096, 4, seed=SELECTION_SEED, device=dev_index, quirks=Q_PAPER)

                res = curriculum_attempts(make_trainer, score, max_attempts=args.curriculum_attempts, accept_touchdown=CURRICULUM_ACCEPT_TOUCHDOWN, comm=comm, rank=rank)
                if rank == 0:
                    hist, att = res["history"], res["attempts"]
                    entered = [a["wall_last_level_by_rule_s"] for a in att if a["wall_last_level_by_rule_s"] is not None]
                    first = att[0]
                    runs.append({"seed": seed, "attempts": len(att), "chosen_attempt": res["chosen"], "accepted": res["accepted"],
                                 # the job's clock when stage 4 was first entered with levels 0-3 promoted by the rule (by whichever attempt got there first);
                                 # no attempt did: when the chosen attempt's level 3 handed over
                                 "wall_to_stage4_s": min(entered) if entered else (att[res["chosen"]]["wall_since_start_s"] - att[res["chosen"]]["wall_train_s"] + hist[3]["wall_since_start_s"] if len(hist) > 3 else None),
                                 "stage4_entered_by_rule": bool(entered), "wall_all_levels_s": res["wall_s"],
                                 "promoted_levels": sum(1 for h in hist if h["promoted"]),
                                 "levels": [{"level": h["level"], "promoted": h["promoted"], "exhausted": h["exhausted"], "restarts": h.get("restarts", 0), "step_backs": h.get("step_backs", 0), "episodes": h["episodes"],
                                             "agent_periods": h["agent_periods"], "wall_s": h["wall_s"],
                                             "population_success_at_promotion": h["success_rate"] if h["promoted"] else None,
                                             "online_success_rate_at_handover": h["success_rate"]} for h in hist],
                                 "attempt_records": att,
                                 "first_attempt": {"promoted_levels": first["promoted_levels"], "stage4_greedy_4096_episodes": greedy(Path(d) / "run0")},
                                 "stage4_greedy_4096_episodes": greedy(Path(d) / f"run{res['chosen']}")})
        if rank != 0:
            return None
        mean = lambda k: sum(r[k] for r in runs) / len(runs)
        pops = [lv["population_success_at_promotion"] for r in runs for lv in r["levels"] if lv["promoted"]]
        # stage 4 is ENTERED by the rule when levels 0-3 were all promoted by it (not handed over by an exhausted budget, pkg/trainer.py:187)
        by_rule = [r for r in runs if r["stage4_entered_by_rule"]]
        firsts = [r["first_attempt"] for r in runs]
        return {"wall_to_stage4_s": mean("wall_to_stage4_s"), "wall_all_levels_s": mean("wall_all_levels_s"),
                "seeds_reaching_stage4_by_rule": len(by_rule), "n_seeds": len(runs),
                "wall_to_stage4_by_rule_s": (sum(r["wall_to_stage4_s"] for r in by_rule) / len(by_rule)) if by_rule else None, "mode": "paper-mode MDP, reference update rule (quirks 0x60), one learning-rate step per agent period (Trainer default)",
                "workload": f"BASELINE configs[3]{' share' if world > 1 or args.curriculum_envs == 32768 else ''}: {args.curriculum_envs} envs per GPU, full curriculum 0 -> 4",
                "envs_per_gpu": args.curriculum_envs, "global_envs": n_global, "episode_budget_per_level": budget, "sync_period": CURRICULUM_SYNC, "trainer_kw": CURRICULUM_KW,
                "seeds": [r["seed"] for r in runs], "promoted_levels_per_seed": [r["promoted_levels"] for r in runs],
                "attempts": {"max": args.curriculum_attempts, "accept_touchdown": CURRICULUM_ACCEPT_TOUCHDOWN, "per_seed": [r["attempts"] for r in runs], "accepted_per_seed": [bool(r["accepted"]) for r in runs],
                             "selection": "attempt j trains from seed + 7919 j; accepted when all five levels were promoted by the rule AND 4 096 greedy landings on the selection batch (seed 977; the figures below fly seed 123) reach accept_touchdown; none accepted: the best seen",
                             "first_attempt": {"promoted_levels_per_seed": [f["promoted_levels"] for f in firsts],
                                               "trained_mean": {k: sum(f["stage4_greedy_4096_episodes"][k] for f in firsts) / len(firsts) for k in ("touchdown_rate", "goal_hold_rate")},
                                               "trained_worst_seed": {k: min(f["stage4_greedy_4096_episodes"][k] for f in firsts) for k in ("touchdown_rate", "goal_hold_rate")}}},
                "level0_promoted_per_seed": [bool(r["levels"][0]["promoted"]) for r in runs],
                "population_success_at_promotion": {"min": min(pops) if pops else None, "mean": sum(pops) / len(pops) if pops else None,
                                                    "note": "success rate of ALL envs' episodes over the chunks holding the most recent >= 100 episodes when the judged envs' deque passed 0.96 / 100"},
                "rule": "deque(100) of the 64 judged envs' episodes in generation order > 0.96 (pkg/trainer.py:218-232) AND population success >= 0.94, or the level's episode budget exhausted (:187); a level above 0 not promoted after 96 episodes per env is started over from the level below (restart_after), a level above 1 that has failed 4 attempts in a row lets the level below be learnt again first (step_back_after = 3, at most 3 times per run, never cascading; a re-learning that is not promoted falls back to the tables the level was promoted with)",
                "stage4_greedy_4096_episodes": {"trained_mean": {k: sum(r["stage4_greedy_4096_episodes"][k] for r in runs) / len(runs) for k in ("touchdown_rate", "goal_hold_rate")},
                                                "trained_worst_seed": {k: min(r["stage4_greedy_4096_episodes"][k] for r in runs) for k in ("touchdown_rate", "goal_hold_rate")},
                                                "reference_assets": greedy(ROOT / "tests" / "golden" / "assets")},
                "runs": runs}
    except Exception as e:  # noqa: BLE001 - the throughput line must survive
        import traceback
        return {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]} if rank == 0 else None


def spawn_ranks(args, child_argv=None) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes (one per GPU; this parent never
    touches the GPU), relay rank 0's JSON line, fail loudly when a rank cannot get its GPU or dies."""
    import socket
    import subprocess
    import tempfile
    import __graft_entry__ as g
    g.build_hip()  # compile once here (no GPU involved) instead of N times under the lock
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    n = args.gpus
    import uuid
    job_id = uuid.uuid4().hex  # per spawn: ranks respawned by this parent on the same port never accept an earlier spawn's bootstrap file (comm.py)
    procs, out0 = [], tempfile.TemporaryFile(mode="w+")
    errs = [sys.stderr] + [tempfile.TemporaryFile(mode="w+") for _ in range(1, n)]  # ranks != 0 are heard only when the job fails
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DQL_COMM_JOB_ID=job_id)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(child_argv or [sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=out0 if r == 0 else errs[r], stderr=errs[r]))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
                break
        time.sleep(0.1)
    if failed is None:
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:  # the others would wait for the dead rank in the communicator: stop exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
        for r in range(1, n):
            errs[r].seek(0)
            tail = errs[r].read()[-2000:]
            if tail.strip():
                print(f"--- rank {r} ---\n{tail}", file=sys.stderr)
        print(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}: fewer than {n} ranks ran, no result (needs {n} GPUs, one per rank)", file=sys.stderr)
        return 1
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    rec = json.loads(lines[-1])
    if rec.get("n_gpus") != n:
        print(f"bench.py: asked for {n} GPUs, the ranks report {rec.get('n_gpus')}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def roofline_block(envs, P, two_axis, randomize_platform, noise, k_ms, dec_per_launch, n_launch, k_pairs_ms=None, valu=None, dtype_name="f32"):
    """`roofline` object of one measurement: achieved = algorithmic bytes per launch / average launch duration (HIP events on the engine's
    stream around the back-to-back launches of the timed region); traffic = HBM bytes per launch from the committed PMC pass of the same
    configuration AND the same kernel sources, else null.  `bound` names the roof that BINDS (VALU issue: the state crosses HBM once per P periods),
    `frac` stays the north_star's figure — SURVEY.md section 8d's algorithmic bytes against the 8 TB/s peak — and `hbm_real_frac` is what the
    kernel really draws (counter bytes / time / peak)."""
    ab = algo_bytes(two_axis, randomize_platform)
    ach = ab * dec_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, note = None, None
    prof, src = committed_profile("traffic")
    key = flavour_key(envs, P, two_axis, randomize_platform, noise)
    if dtype_name != "f32":
        ab *= 2  # the same 40 / 41 / 50 words of state per env, 8 bytes each
        ach *= 2
        note = "no PMC pass of the float64 kernel is committed"
    elif prof is None:
        note = src
    elif key not in prof["configs"]:
        note = f"profiles/{src} has no pass of configuration {key}"
    else:
        traffic = prof["configs"][key]["hbm_bytes_per_env_step"] * dec_per_launch
        note = (f"profiles/{src} [{key}], same kernel sources ({prof['source_sha16']}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH doubled per the "
                "gfx950 correction), HBM bytes per env-step x env-steps per launch of this run")
    out = {"bound": "valu_issue", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": note,
           "frac_is": "hbm_algorithmic", "hbm_real_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and k_ms > 0) else None,
           "valu_issue_frac": (valu or {}).get("frac_at_2p4_ghz"),
           "kernel": "k_step", "kernel_avg_ms": k_ms, "kernel_launches_timed": n_launch, "agent_periods_per_launch": P,
           "algorithmic_bytes_per_env_step": ab, "env_steps_per_launch": dec_per_launch,
           "note": "bound = the roof that binds: wave64 VALU issue (~22 physics ticks per 328 B of state; the env stays in registers for P periods, so the real "
                   "HBM draw is hbm_real_frac).  achieved / peak / frac = the north_star's HBM figure: SURVEY.md 8d algorithmic bytes per env-step x env-steps per "
                   "launch / launch duration against 8 TB/s; valu_issue_frac = VALU instructions x 2 cycles per SIMD / (duration x 2.4 GHz)"}
    if k_pairs_ms is not None:
        out["kernel_avg_ms_event_pairs"] = k_pairs_ms
    return out


def single_gpu_block(tag, what, envs, two_axis, randomize_platform, noise, dtype, dtype_name, P, eps, steps, warmup, device, block=0):
    """One single-GPU throughput measurement with its own roofline (secondary blocks of the line: small_batch, large_batch)."""
    from dql_multirotor_landing_amd.config import DqlConfig
    from dql_multirotor_landing_amd.engine import Engine
    cfg = DqlConfig(dtype=dtype, two_axis=two_axis, per_env_platform=randomize_platform, fold_per_step=1,
                    noise_pos_sd=0.25 if noise else 0.0, noise_vel_sd=0.1 if noise else 0.0)
    eng = Engine(cfg, envs, seed=42, device=device)
    eng.set_option("block", block)
    eng.set_option("periods_per_launch", P)
    eng.train_steps(warmup, eps); eng.sync()
    s0 = eng.stats(); eng.timer_start(); t0 = time.perf_counter()
    eng.train_steps(steps, eps)
    dev_ms = eng.timer_stop(); wall = time.perf_counter() - t0
    s1 = eng.stats()
    eng.close()
    dec = s1["decisions"] - s0["decisions"]
    n_launch = -(-steps // P)
    k_ms = dev_ms / n_launch
    v = valu_issue(envs, P, two_axis, randomize_platform, noise, dtype_name, k_ms, steps / n_launch)
    out = {"workload": f"{tag}: {what}", "envs": envs, "dtype": dtype_name, "value": dec / wall, "unit": "env-steps/s", "steps": steps, "warmup": warmup, "ms_per_step": wall * 1e3 / steps,
           "device_ms_per_step": dev_ms / steps, "periods_per_launch": P,
           "roofline": roofline_block(envs, P, two_axis, randomize_platform, noise, k_ms, dec / n_launch, n_launch, valu=v, dtype_name=dtype_name)}
    if v:
        out["valu_issue"] = v
    return out


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 7000  # bytes: the driver keeps the last 8 KB of stdout; round 3's 26 KB line was cut and did not parse


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(full: dict) -> dict:
    """The ONE line bench.py prints: the contract's keys + roofline + cpu_baseline + the headline figures of every leg, <= LINE_LIMIT bytes
    whatever the number of curriculum seeds or ranks.  Everything else (per-seed runs, per-level records, notes, sources, the other sync
    periods) goes to DETAIL_FILE next to this script, which the line names."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "preroll_steps", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    if full.get("repeats"):
        out["repeats"] = _pick(full["repeats"], ("n", "value_min", "value_max", "ms_per_step_min", "ms_per_step_max", "statistic"))
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("workload", "baseline_config", "envs_per_gpu", "global_envs", "sync_period", "periods_per_launch", "parallelism", "exchange_rehearsal",
                                "algorithmic_bytes_per_env_step", "library_source_sha16"))
    out["config"]["workload"] = str(cfg.get("workload", ""))[:160]
    if not out["config"].get("exchange_rehearsal"):
        out["config"].pop("exchange_rehearsal", None)
    out["env_steps"] = full.get("env_steps")
    roof_keys = ("bound", "achieved", "peak", "unit", "frac", "frac_is", "hbm_real_frac", "valu_issue_frac", "traffic", "kernel", "kernel_avg_ms", "kernel_launches_timed",
                 "agent_periods_per_launch", "algorithmic_bytes_per_env_step", "env_steps_per_launch")
    out["roofline"] = _pick(full.get("roofline"), roof_keys)
    issue_keys = ("valu_instr_per_env_wave_per_period", "env_waves_per_simd", "cycles_per_instr", "frac_at_2p4_ghz", "measured_clock_ghz", "frac_at_measured_clock")
    if isinstance(full.get("valu_issue"), dict) and "frac_at_2p4_ghz" in full["valu_issue"]:
        out["valu_issue"] = _pick(full["valu_issue"], issue_keys)
    for blk in ("small_batch", "large_batch"):
        b = full.get(blk)
        if b:
            out[blk] = {**_pick(b, ("envs", "value", "ms_per_step")), "roofline_frac": b["roofline"]["frac"], "kernel_avg_ms": b["roofline"]["kernel_avg_ms"],
                        "traffic": b["roofline"].get("traffic")}
            if isinstance(b.get("valu_issue"), dict) and "frac_at_2p4_ghz" in b["valu_issue"]:
                out[blk]["valu_issue_frac_at_2p4_ghz"] = b["valu_issue"]["frac_at_2p4_ghz"]
                out[blk]["valu_issue_frac_at_measured_clock"] = b["valu_issue"]["frac_at_measured_clock"]
    if full.get("f64"):
        b = full["f64"]
        out["f64"] = {**_pick(b, ("envs", "value", "ms_per_step")), "kernel_avg_ms": b["roofline"]["kernel_avg_ms"],
                      "roofline": _pick(b["roofline"], ("bound", "achieved", "peak", "unit", "frac", "hbm_real_frac", "traffic"))}
    if full.get("eps_0p1"):
        out["eps_0p1"] = _pick(full["eps_0p1"], ("eps", "value", "ms_per_step", "kernel_avg_ms", "roofline_frac", "value_min", "value_max"))
    if full.get("sync"):
        out["sync"] = _pick(full["sync"], ("exchange_name", "sync_period", "ms_per_step", "ms_per_step_no_exchange", "sync_ms_per_step", "exchange_device_ms", "exchanges_timed",
                                           "staleness_bound_periods", "replicas_identical", "p2p_failed"))
        for other in ("p2p", "rccl"):
            if isinstance(full["sync"].get(other), dict):
                o = full["sync"][other]
                out["sync"][other] = _pick(o, ("value", "ms_per_step", "sync_ms_per_step", "exchange_device_ms", "replicas_identical"))
                if "skipped" in o:
                    out["sync"][other]["skipped"] = str(o["skipped"])[:200]
        out["sync_ms_per_step"] = full["sync"].get("sync_ms_per_step")
    cur = full.get("curriculum")
    if cur is not None:
        if "error" in cur:
            out["curriculum"] = {"error": str(cur["error"])[:300]}
        else:
            pop = cur.get("population_success_at_promotion") or {}
            out["curriculum"] = {**_pick(cur, ("wall_to_stage4_s", "wall_all_levels_s", "seeds_reaching_stage4_by_rule", "n_seeds", "wall_to_stage4_by_rule_s", "envs_per_gpu",
                                               "global_envs", "sync_period", "seeds", "promoted_levels_per_seed")),
                                 "population_success_at_promotion": _pick(pop, ("min", "mean")),
                                 "attempts": _pick(cur.get("attempts"), ("max", "accept_touchdown", "per_seed", "accepted_per_seed", "first_attempt")),
                                 "stage4_greedy_4096_episodes": _pick(cur.get("stage4_greedy_4096_episodes"), ("trained_mean", "trained_worst_seed", "reference_assets"))}
            out.update(_pick(full, ("promoted_levels", "goal_hold_rate", "touchdown_rate", "wall_to_stage4_s")))
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "single_thread_value"))
        out["cpu_baseline"]["sample"] = str(out["cpu_baseline"].get("sample", ""))[:200]
    out["reference_quoted"] = _pick(full.get("reference_quoted"), ("reference+gazebo_env_steps_per_s", "realtime_ceiling", "reference_python_mdp+agent_steps_per_s"))
    out["detail_file"] = DETAIL_FILE
    # last resort, never expected: shed optional blocks rather than print a line the driver cannot parse
    for k in ("reference_quoted", "large_batch", "small_batch", "eps_0p1", "valu_issue", "f64", "sync", "curriculum"):
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        out.pop(k, None)
    return out


def emit(full: dict) -> str:
    """write the full record to DETAIL_FILE (next to this script; best effort) and return the compact line"""
    try:
        tmp = ROOT / (DETAIL_FILE + f".{os.getpid()}.tmp")
        tmp.write_text(json.dumps(full, indent=1))
        os.replace(tmp, ROOT / DETAIL_FILE)
    except OSError as e:
        print(f"bench.py: could not write {DETAIL_FILE}: {e}", file=sys.stderr)
    return json.dumps(compact_line(full))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--preroll", type=int, default=512,
                    help="untimed agent periods run on the engine BEFORE the W warm-up periods: a GPU that was idle while the process started runs its first "
                         "millisecond at idle clocks, and W = 5 periods (0.2 ms) do not reach the sustained clock (measured: 47.8 vs 35.3 us per period); reported in the line")
    ap.add_argument("--config", type=int, default=4, choices=[1, 2, 3, 4], help="BASELINE.json configs[k] preset of the throughput leg (per-GPU share); default 4 at every N")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (weak scaling); overrides the preset")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--eps", type=float, default=1.0)
    ap.add_argument("--repeats", type=int, default=7,
                    help="the timed region (exactly K steps between barrier + device sync) is run this many times back to back; `value` / `ms_per_step` / the roofline are the "
                         "MEDIAN repetition's, min and max next to them (a single 20-step region is 0.5 ms: one draw says little)")
    ap.add_argument("--sync-period", type=int, default=16,
                    help="agent periods between table exchanges (N > 1).  Default 16 = periods per launch: the smallest window that costs no extra launch boundary; "
                         "a rank then acts on tables that miss at most the other ranks' last 16 + 16 (one-launch fold delay) + 16 (window in flight) = 48 periods")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--periods-per-launch", type=int, default=16,
                    help="agent periods per kernel launch (engine option; 1 = one launch per period).  16: the launch boundary (state round trip through HBM, "
                         "dispatch ramp and tail: 33 us at 131 072 envs) is paid once per 16 periods (32.2 -> 29.8 us per period against 8)")
    ap.add_argument("--two-axis", type=int, default=None, help="1 = joint x+y MDP (overrides the preset)")
    ap.add_argument("--randomize-platform", type=int, default=None, help="1 = per-env platform amplitude / speed (overrides the preset)")
    ap.add_argument("--noise", type=int, default=None, help="1 = observation noise 0.25 m / 0.1 m/s + Kalman R = 0.1^2 (overrides the preset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "p2p"],
                    help="table exchange between ranks: RCCL all-reduce (default) or the one-shot peer-to-peer push + local sum (dql_p2p_*: opt-in, "
                         "functionally tested with ranks sharing one GPU, not yet measured across GPUs)")
    ap.add_argument("--exchange-rehearsal", action="store_true",
                    help="one rank, but through every code path of a multi-rank run (RCCL communicator of world size 1, window exchange, sync legs, sharded curriculum): a rehearsal on a 1-GPU box, flagged in the line, not a measurement")
    ap.add_argument("--cpu-steps", type=int, default=600, help="agent periods of the single-thread CPU sample (x cores for the all-core sample): ~5 s + ~7 s")
    ap.add_argument("--small-envs", type=int, default=4096, help="secondary single-GPU block at BASELINE configs[1] (0 = skip)")
    ap.add_argument("--large-envs", type=int, default=1048576, help="secondary single-GPU block at a chip-filling batch (0 = skip)")
    ap.add_argument("--no-f64-block", dest="f64_block", action="store_false", help="skip the float64 block of the single-GPU line (same workload in the reference's precision)")
    ap.add_argument("--no-curriculum", action="store_true", help="skip the wall-clock-to-stage-4 leg")
    ap.add_argument("--curriculum-envs", type=int, default=32768, help="envs per GPU of the curriculum leg (BASELINE configs[3]: 262 144 / 8)")
    ap.add_argument("--curriculum-seeds", type=int, default=None,
                    help="how many of the twelve seeds to run; default 12 on one GPU, 2 with several (the scaling sweep re-runs the leg at every N: its throughput leg is the same at N = 1 either way)")
    ap.add_argument("--curriculum-attempts", type=int, default=CURRICULUM_ATTEMPTS,
                    help="whole curricula a seed may train (seed + 7919 j) until one has all five levels by the rule and lands >= 0.875 on the selection batch; 1 = one run per seed, as rounds 1-4")
    ap.add_argument("--curriculum-budget", type=int, default=50000, help="episodes per level before the next level starts (reference: 50000); at least 768 per env")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.curriculum_seeds is None:
        args.curriculum_seeds = 12 if args.gpus == 1 else 2
    pre = PRESETS[args.config]
    for k in ("envs", "two_axis", "randomize_platform", "noise"):
        if getattr(args, k) is None:
            setattr(args, k, pre[k])
    custom = any(getattr(args, k) != pre[k] for k in ("envs", "two_axis", "randomize_platform", "noise"))

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # by torch.distributed.run, or by spawn_ranks below
    if not launched and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) if launched else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)

    import __graft_entry__ as g
    g.build_hip()
    from dql_multirotor_landing_amd.comm import RcclComm
    from dql_multirotor_landing_amd.config import DqlConfig, F32, F64
    from dql_multirotor_landing_amd.dist import P2PWindowReducer, RcclWindowReducer, ShardedRunner, median_repetition, replicas_identical, timed_region
    from dql_multirotor_landing_amd.engine import Engine

    dtype = F32 if args.dtype == "f32" else F64
    dev_index = local_rank
    comm = None
    multi = world > 1 or args.exchange_rehearsal  # the exchange path (always with several ranks)
    if args.exchange_rehearsal and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        comm = RcclComm(0, 1, dev_index)
    if world > 1:
        try:
            comm = RcclComm.from_env(dev_index)  # one GPU per rank or an error: never a silent 1-GPU measurement
        except Exception as e:  # noqa: BLE001
            print(f"bench.py rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
            sys.exit(3)

    # fold_per_step = 1: the Trainer's default table update (one learning-rate step per agent period towards the period's mean
    # target) — the throughput leg runs the update rule the curriculum leg trains with
    cfg_kw = dict(two_axis=args.two_axis, per_env_platform=args.randomize_platform, fold_per_step=1,
                  noise_pos_sd=0.25 if args.noise else 0.0, noise_vel_sd=0.1 if args.noise else 0.0)
    cfg = DqlConfig(dtype=dtype, working_curriculum_step=0, **cfg_kw)
    eng = Engine(cfg, args.envs, seed=42, device=dev_index, env_id_offset=rank * args.envs)
    eng.set_option("block", args.block)
    # any --steps K works: the engine cuts K periods into launches of at most P (the last one shorter)
    if not 1 <= args.periods_per_launch <= 32:
        print("bench.py: --periods-per-launch must be in 1..32", file=sys.stderr)
        sys.exit(2)
    eng.set_option("periods_per_launch", args.periods_per_launch)
    # BOTH exchanges are set up on the one engine (round 5): the run's `value` is timed with --exchange (default RCCL), the other one gets its own
    # timed leg in the same run (sync.<name> block) — one `bench.py --gpus 8` answers which is faster.  The peer-to-peer set-up may be refused
    # (HIP IPC between these devices): then its leg is reported as skipped, with the reason; its wait is bounded (p2p_spin_limit), never a hang.
    reducers, p2p_setup_error = {}, None
    if multi:
        reducers["rccl"] = RcclWindowReducer(eng, comm)
        try:
            eng.set_option("p2p_spin_limit", 4000000)  # ~ seconds: a peer that never shows up ends the leg, not the job's time limit
            reducers["p2p"] = P2PWindowReducer(eng, rank, world)
        except Exception as e:  # noqa: BLE001
            p2p_setup_error = f"{type(e).__name__}: {e}"[:300]
        ok = float("p2p" in reducers)
        if comm and world > 1:  # on every rank or on none
            ok = float(-comm.all_reduce_max([-ok])[0])
        if not ok:
            reducers.pop("p2p", None)
            p2p_setup_error = p2p_setup_error or "the peer-to-peer exchange could not be set up on another rank"
        if args.exchange == "p2p" and "p2p" not in reducers:
            print(f"bench.py rank {rank}: --exchange p2p: {p2p_setup_error}", file=sys.stderr)
            sys.exit(3)
    reducer = reducers.get(args.exchange)

    def barrier():
        eng.sync()
        if comm:
            comm.barrier()
            eng.sync()

    def timed(sync_period, steps, warmup, final_exchange=True, red="default", reps=1, eps=None):
        """dist.timed_region on this run's engine / communicator: W untimed + `reps` x exactly `steps` timed agent periods, barrier + device sync on
        both sides of every repetition, MAX over ranks / SUM of env-steps; None when the device part failed on any rank (every rank still took part
        in every collective)"""
        return timed_region(eng, comm, reducer if red == "default" else red, sync_period, steps, warmup, args.eps if eps is None else eps, reps=reps,
                            final_exchange=final_exchange, log=lambda m: print(f"bench.py rank {rank}: {m}", file=sys.stderr))

    def median_rep(rs):
        return median_repetition(rs, args.steps)

    # pre-roll + W warm-up periods, all untimed, all on the run's own table schedule (exchanges included with several ranks)
    reps = timed(args.sync_period, args.steps, args.preroll + args.warmup, reps=max(1, args.repeats))
    if reps is None:
        print(f"bench.py rank {rank}: the timed region failed", file=sys.stderr)
        sys.exit(4)
    (wall, decisions, dev_ms), spread = median_rep(reps)
    sync_info = None
    if multi:
        identical = replicas_identical(eng, comm)   # the timed region ended on exchanged tables: every rank must hold the same three tables now
        EXCHANGE_TEXT = {"p2p": "flush + push of 11 340 int64 words into every rank's exchange buffer (HIP IPC, uncached) + flags + local sum + fold, on the engine's stream",
                         "rccl": "flush + ncclAllReduce(ncclInt64, ncclSum, 11 340 words = 90 720 B) + fold, on the engine's stream"}
        # the exchange's price: same region without exchanges (one window, folded after the clock stops: NOT a valid training
        # schedule, a yardstick), and at sync_period 2 (the regime in which the run does not depend on the number of ranks)
        r_none = timed(args.steps + args.warmup + 1, args.steps, 0, final_exchange=False, reps=3)
        (w_none, d_none, _), _sp = median_rep(r_none)
        others = {}
        for sp in (2, 16, 32):
            if sp != args.sync_period:
                (w_sp, d_sp, _), _s = median_rep(timed(sp, args.steps, 0, reps=3))
                others[f"sync_period_{sp}"] = {"value": d_sp / w_sp, "ms_per_step": w_sp * 1e3 / args.steps, "sync_ms_per_step": (w_sp - w_none) * 1e3 / args.steps}

        def exchange_device_ms(red):
            eng.kernel_timer(True)
            r2 = ShardedRunner(eng, red, sync_period=args.sync_period); r2.train_steps(4 * args.sync_period, args.eps); r2.sync()
            ms, n_ = eng.sync_time_ms()
            eng.kernel_timer(False)
            return ms, n_
        sync_dev_ms, n_sync = exchange_device_ms(reducer)
        sync_info = {"exchange_name": args.exchange, "sync_period": args.sync_period, "ms_per_step": wall * 1e3 / args.steps, "ms_per_step_no_exchange": w_none * 1e3 / args.steps,
                     "sync_ms_per_step": (wall - w_none) * 1e3 / args.steps, "exchange_device_ms": sync_dev_ms, "exchanges_timed": n_sync,
                     "staleness_bound_periods": args.sync_period + 2 * max(args.sync_period, args.periods_per_launch),
                     "staleness_note": "a rank acts on tables that hold every rank's updates older than this many agent periods: the window in flight, the launch whose accumulators are being folded, and the launch in progress",
                     "exchange": EXCHANGE_TEXT[args.exchange], "replicas_identical": identical, **others}
        if not identical:
            print(f"bench.py rank {rank}: the table replicas DIFFER between ranks after the final exchange ({args.exchange}): the run is invalid", file=sys.stderr)
            if comm:
                comm.barrier(); comm.close()
            sys.exit(5)

    # SURVEY.md section 8d config 2 asks "eps = 1.0 then 0.1": the same engine carries on at eps = 0.1 on the tables it has learnt so far (longer
    # episodes, fewer resets, greedy choices from the table rows): a second, shorter timed leg
    eps01 = None
    if not multi and abs(args.eps - 0.1) > 1e-12:
        r01 = timed(args.sync_period, args.steps, args.warmup, reps=3, eps=0.1)
        if r01 is not None:
            (w01, d01, ms01), sp01 = median_rep(r01)
            eps01 = {"eps": 0.1, "value": d01 / w01, "ms_per_step": w01 * 1e3 / args.steps, "kernel_avg_ms": ms01 / -(-args.steps // args.periods_per_launch),
                     "value_min": sp01["value_min"], "value_max": sp01["value_max"], "env_steps": d01,
                     "note": "same engine, right after the eps = 1.0 region, training continued at eps = 0.1 on the tables learnt so far"}

    # Average launch duration of the fused step kernel, HIP events on the engine's stream.  One rank: the step kernel is the
    # only kernel between the two events of the timed region (K launches back to back), so duration = region / K — the figure
    # rocprofv3's kernel summary of the same command reports.  Event PAIRS around single
    # launches (second figure) add the event records and an idle boundary per launch; with several ranks the timed region also
    # holds the exchange kernels, so there the pairs are the per-launch figure.
    P = args.periods_per_launch
    s1 = eng.stats()
    eng.kernel_timer(True)
    eng.train_steps(P * min(200, max(20, args.steps // (10 * P))), args.eps)
    k_pairs_ms, k_n = eng.kernel_time_ms()
    eng.kernel_timer(False)
    s2 = eng.stats()
    dec_per_launch = P * (s2["decisions"] - s1["decisions"]) / max(1, s2["agent_steps"] - s1["agent_steps"])  # one launch = P agent periods
    n_launch = -(-args.steps // P)  # ceil: the last launch of the timed region may hold fewer than P periods
    k_ms = dev_ms / n_launch if not multi else k_pairs_ms
    if not multi:
        dec_per_launch = decisions / n_launch  # average over the launches of the timed region (kernel_avg_ms is their average duration)
    if multi:  # last use of the engine: a peer-to-peer leg that fails leaves dql_stats_get raising DQL_EPEER from then on
        # the OTHER exchange, same engine, same schedule, same run (its failure ends its leg, not the run: the line's value is already measured)
        other = "p2p" if args.exchange == "rccl" else "rccl"
        if other not in reducers:
            sync_info[other] = {"skipped": p2p_setup_error}
        else:
            r_o = timed(args.sync_period, args.steps, args.warmup, red=reducers[other], reps=max(3, args.repeats // 2))
            if r_o is None:
                sync_info[other] = {"skipped": "the exchange failed inside its timed leg (a peer was not seen within p2p_spin_limit polls); see stderr"}
            else:
                (w_o, d_o, _), sp_o = median_rep(r_o)
                ident_o = replicas_identical(eng, comm)
                ms_o, n_o = exchange_device_ms(reducers[other])
                sync_info[other] = {"value": d_o / w_o, "ms_per_step": w_o * 1e3 / args.steps, "sync_ms_per_step": (w_o - w_none) * 1e3 / args.steps, "exchange_device_ms": ms_o,
                                    "exchanges_timed": n_o, "replicas_identical": ident_o, "value_min": sp_o["value_min"], "value_max": sp_o["value_max"], "exchange": EXCHANGE_TEXT[other]}
        if "p2p" in reducers:
            try:
                sync_info["p2p_failed"] = eng.p2p_failed()
            except Exception:  # noqa: BLE001
                sync_info["p2p_failed"] = True
    eng.close()

    curriculum = None
    if not args.no_curriculum:
        curriculum = curriculum_leg(args, comm, world, rank, dev_index, dtype)

    if rank == 0:
        value = decisions / wall
        ab = algo_bytes(args.two_axis, args.randomize_platform)
        tag = (pre["tag"] if not custom else f"custom (preset {pre['tag']} with overrides)")
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "preroll_steps": args.preroll, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{tag}: " + (pre["short"] if not custom else f"{args.envs} envs per GPU, two_axis {args.two_axis}, per-env platform {args.randomize_platform}, noise {args.noise}"),
                       "workload_long": f"{tag}: " + (pre["what"] if not custom else f"{args.envs} envs per GPU, two_axis {args.two_axis}, per-env platform {args.randomize_platform}, noise {args.noise}") +
                                   f"; eps {args.eps}, ONE fused kernel per {P} agent period(s) (env steps + table fold in writer workgroups), int64 LDS/global accumulators",
                       "baseline_config": args.config, "envs_per_gpu": args.envs, "global_envs": args.envs * world, "sync_period": args.sync_period if multi else 1,
                       "exchange_rehearsal": bool(args.exchange_rehearsal), "periods_per_launch": P, "fold_per_step": 1,
                       "parallelism": f"env-shard x{world}" + ((", one-shot peer-to-peer window exchange (libdql_hip.so, HIP IPC)" if args.exchange == "p2p" else ", RCCL int64 window all-reduce (libdql_hip.so, no PyTorch)") if multi else ""), "block": args.block,
                       "two_axis": args.two_axis, "randomize_platform": args.randomize_platform, "noise": args.noise, "algorithmic_bytes_per_env_step": ab,
                       "library_source_sha16": lib_source_sha16()},
            "env_steps": decisions, "device_ms_per_step": dev_ms / args.steps, "repeats": spread,
            "reference_quoted": {"reference+gazebo_env_steps_per_s": 20.18, "realtime_ceiling": 22.92,
                                 "reference_python_mdp+agent_us_per_step": 69.6, "reference_python_mdp+agent_steps_per_s": 14.4e3,
                                 "source": "BASELINE.md section 2 (artefact-derived / measured in the survey container on 1 core; not re-measurable on the GPU box)"},
        }
        if sync_info:
            out["sync"] = sync_info
            out["sync_ms_per_step"] = sync_info["sync_ms_per_step"]
        valu = valu_issue(args.envs, P, args.two_axis, args.randomize_platform, args.noise, args.dtype, k_ms, args.steps / n_launch if not multi else P)
        if valu:
            out["valu_issue"] = valu
        out["roofline"] = roofline_block(args.envs, P, args.two_axis, args.randomize_platform, args.noise, k_ms, dec_per_launch, n_launch if not multi else k_n, k_pairs_ms, valu)
        if eps01:
            eps01["roofline_frac"] = ab * eps01["env_steps"] / -(-args.steps // P) / (eps01["kernel_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            out["eps_0p1"] = eps01
        if not multi:
            # secondary single-GPU blocks, each with its own roofline from stream events of THIS run
            if args.small_envs > 0 and not (args.envs == args.small_envs and not custom and args.config == 1):
                p1 = PRESETS[1]
                out["small_batch"] = single_gpu_block(p1["tag"], p1["what"], args.small_envs, 0, 0, 0, dtype, args.dtype, P, args.eps, max(args.steps, 400), max(args.warmup, 40), dev_index)
            # the precision at which parity with the reference is BIT-EXACT (pkg/mdp.py:11-32 and everything behind it is float64): same workload, float64 kernel
            if args.dtype == "f32" and args.f64_block:
                out["f64"] = single_gpu_block(tag + " in float64 (the reference's precision: reference == oracle == kernel bit for bit)", pre["short"] if not custom else "custom", args.envs,
                                              args.two_axis, args.randomize_platform, args.noise, F64, "f64", P, args.eps, 10 * P, 2 * P, dev_index)
            if args.large_envs > 0 and args.envs != args.large_envs:
                out["large_batch"] = single_gpu_block("chip-filling batch (= configs[4] on ONE GPU)", "1 048 576 envs, per-env randomised platforms + observation noise" if args.large_envs == 1048576 else f"{args.large_envs} envs, per-env randomised platforms + observation noise",
                                                      args.large_envs, 0, 1, 1, dtype, args.dtype, P, args.eps, 40 * P, 5 * P, dev_index)
        if curriculum is not None:
            out["curriculum"] = curriculum
            if "error" not in curriculum:  # the figures the stage-4 check is read from, at the top level
                out["promoted_levels"] = curriculum["promoted_levels_per_seed"]
                out["goal_hold_rate"] = curriculum["stage4_greedy_4096_episodes"]["trained_mean"]["goal_hold_rate"]
                out["touchdown_rate"] = curriculum["stage4_greedy_4096_episodes"]["trained_mean"]["touchdown_rate"]
                out["wall_to_stage4_s"] = curriculum["wall_to_stage4_s"]
        if not args.no_cpu_baseline and not multi:
            out["cpu_baseline"] = cpu_baseline(min(args.envs, 4096), args.cpu_steps, dtype, {k: v for k, v in cfg_kw.items()})
        print(emit(out), flush=True)
    if comm:
        comm.barrier()
        comm.close()


def committed_clock(envs: int, cfg4: bool):
    """shader clock measured INSIDE the step kernel at this batch size (tools/exp_phase_clock.py, diagnostic build: s_memtime cycles of a
    launch / its HIP-event duration), newest committed profiles/r*_phase_clock.jsonl; (GHz, file) or (None, why)"""
    for f in sorted((ROOT / "profiles").glob("r*_phase_clock.jsonl"), reverse=True):
        rows = []
        for ln in f.read_text().splitlines():
            try:
                rows.append(json.loads(ln))
            except ValueError:
                pass
        rows = [r for r in rows if r.get("envs") == envs and 1.2 < r.get("implied_clock_ghz", 0) <= 2.6]  # (an oversubscribed batch's waves do not all run at once: cycles / time is no clock there — 0.5 "GHz" at 1 M envs)
        rows.sort(key=lambda r: (r.get("flavour") == "cfg4") == cfg4, reverse=True)
        if rows:
            return rows[0]["implied_clock_ghz"], f.name
    return None, "no committed profiles/r*_phase_clock.jsonl row for this batch size"


def valu_issue(envs, P, two_axis, randomize_platform, noise, dtype_name, k_ms, periods_per_launch_avg):
    """The roof that binds the fused step: wave64 VALU issue.  ONE price, MI355X_MICROARCH.md's table row (v_fma_f32 wave64: 2 cycles per
    instruction per SIMD; a wave alone on its SIMD: 4), applied to the instruction count per env wave of the committed PMC pass of THIS
    library's sources (profiles/r*_pmc_sq_summary.json, stamped):
        valu_issue_frac = instructions per SIMD per launch x cycles / (kernel time x clock)
    at the guide's 2.4 GHz and at the clock measured inside the kernel for this batch size (the chip sustains 1.9-2.0 GHz under a dense VALU
    stream).  A fraction of what the SIMDs could issue: never above 1."""
    if two_axis or dtype_name != "f32" or not k_ms or k_ms <= 0:
        return None
    pm, src = committed_profile("pmc_sq_summary")
    if pm is None:
        return {"note": src}
    passes = pm["configs"]
    key = flavour_key(envs, P, two_axis, randomize_platform, noise)
    if key not in passes:  # the pass of this env count and flavour at another periods-per-launch (the count per period moves by < 2 % with it)
        suffix = key.split(f"_p{P}", 1)[1]
        same = [k for k in passes if k.split("_p")[0] == str(envs) and k.split("_p", 1)[1].lstrip("0123456789") == suffix]
        if not same:
            return {"note": f"profiles/{src} has no pass of configuration {key}"}
        key = same[0]
    per_period = passes[key]["SQ_INSTS_VALU_per_env_wave_per_period"]
    waves = (envs + 63) // 64
    lone = waves <= 1024
    cycles = 4 if lone else 2
    instr_per_simd = per_period * periods_per_launch_avg * max(1.0, waves / 1024.0)
    clock, clock_src = committed_clock(envs, bool(randomize_platform and noise))
    frac = lambda ghz: instr_per_simd * cycles / (k_ms * 1e-3 * ghz * 1e9)
    return {"valu_instr_per_env_wave_per_period": per_period, "env_waves_per_simd": waves / 1024.0, "cycles_per_instr": cycles,
            "frac_at_2p4_ghz": frac(2.4), "measured_clock_ghz": clock, "frac_at_measured_clock": frac(clock) if clock else None,
            "source": f"profiles/{src} [{key}]" + (f", profiles/{clock_src}" if clock else f"; {clock_src}")}


if __name__ == "__main__":
    main()
