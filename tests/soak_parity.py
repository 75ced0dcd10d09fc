#!/usr/bin/env python3
"""One-off randomized parity soak: HIP engine vs CPU oracle over random points of the feature matrix (dtype, level, quirk
set, axes, trajectory, per-env platform, observation noise, fold semantics, windowed exchange, block size, tick layout, periods per
launch), every field and
table compared bit for bit after every chunk.  The regular suite pins chosen points; this sweeps the space once.

    python tests/soak_parity.py [N_CONFIGS=24] [SEED=0]
"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32, F64, TRAJ_EIGHT
from dql_multirotor_landing_amd.engine import Engine
from oracle.oracle import Oracle

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for k in range(n_cfg):
    kw = dict(dtype=int(rng.choice([F32, F32, F64])), working_curriculum_step=int(rng.integers(0, 5)), quirks=int(rng.choice([0x7F, 0x00, 0x40, 0x77, 0x60, int(rng.integers(0, 128))])),
              two_axis=int(rng.random() < 0.3), fold_per_step=int(rng.random() < 0.2), t_max=float(rng.choice([20.0, 4.0, 2.0])))
    if rng.random() < 0.3: kw["trajectory"] = TRAJ_EIGHT
    if rng.random() < 0.3: kw.update(per_env_platform=1)
    if rng.random() < 0.3: kw.update(noise_pos_sd=0.25, noise_vel_sd=0.1)
    if rng.random() < 0.3: kw.update(init_uniform=1)
    n = int(rng.choice([1, 63, 64, 65, 200, 512, 700, 3000]))
    seed = int(rng.integers(0, 2**31))
    windowed = rng.random() < 0.3
    block = int(rng.choice([0, 0, 64, 128, 256, 512])) if kw["dtype"] == F32 else int(rng.choice([0, 0, 64, 128, 256]))
    tick = int(rng.integers(0, 5)) if kw["dtype"] == F32 else int(rng.integers(0, 4))   # 4 = literal constants: float32 + reference vehicle (the default config)
    ppl = int(rng.choice([1, 1, 2, 3, 4, 8, 13, 16, 24, 32]))
    eng = Engine(DqlConfig(**kw), n, seed=seed); orc = Oracle(DqlConfig(**kw), n, seed=seed, n_threads=8)
    eng.set_option("block", block); eng.set_option("tick", tick)
    eng.set_option("fair_prio", int(rng.integers(-1, 2)))   # round 5: issue-priority alternation (scheduling only: must change nothing)
    eng.set_option("periods_per_launch", ppl); orc.set_option("periods_per_launch", ppl)
    if windowed:
        eng.set_windowed(True); orc.set_windowed(True)
    ok = True
    for chunk in range(6):
        steps, eps = int(rng.integers(1, 60)), float(rng.choice([1.0, 0.5, 0.05, 0.0]))
        mode = rng.random()
        if mode < 0.75:
            eng.train_steps(steps, eps); orc.train_steps(steps, eps)
        elif mode < 0.9:
            eng.eval_steps(steps); orc.eval_steps(steps)
        else:
            act = rng.integers(0, 3, size=n).astype(np.uint8)
            if kw["two_axis"]: act = (act | (rng.integers(0, 3, size=n).astype(np.uint8) << 2)).astype(np.uint8)
            eng.step(act); orc.step(act)
        if windowed and rng.random() < 0.5:
            eng.flush(); orc.flush()
            assert np.array_equal(eng.get_accum(), orc.get_accum())
            eng.apply_accum(); orc.apply_accum()
        er, ei = eng.get_fields(); o_r, o_i = orc.get_fields()
        qa, qb, cnt = eng.get_tables()
        ok = (np.array_equal(ei, o_i) and np.array_equal(er, o_r, equal_nan=True) and np.array_equal(qa.ravel(), orc.qa) and np.array_equal(qb.ravel(), orc.qb)
              and np.array_equal(cnt.ravel(), orc.count))
        if not ok:
            break
    se, so = eng.stats(), orc.stats_dict()
    ok = ok and se["decisions"] == so["decisions"] and list(se["by_code"].values()) == so["by_code"] and se["reward_sum"] == so["reward_sum"]
    bad += not ok
    print(json.dumps({"config": k, "ok": bool(ok), "n": n, "seed": seed, "windowed": bool(windowed), "block": block, "tick": tick, "periods_per_launch": ppl, "episodes": se["episodes"], **{a: (float(b) if isinstance(b, float) else int(b)) for a, b in kw.items()}}), flush=True)
    eng.close()
print(json.dumps({"configs": n_cfg, "mismatching": bad}))
sys.exit(1 if bad else 0)
