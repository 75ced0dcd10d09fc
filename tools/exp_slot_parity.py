#!/usr/bin/env python3
"""Do the two env waves that share a SIMD sit in hardware wave slots of different parity?  (k_step's issue-priority alternation takes its role from
HW_ID.wave_id & 1.)  Diagnostic build -DDQL_WAVE_CLOCK=8 through DQL_LIB_PATH: per wave HW_ID | XCC_ID << 32.
    tools/ab_build.sh clock8 -DDQL_WAVE_CLOCK=8; DQL_LIB_PATH=.../libdql_hip_clock8.so python tools/exp_slot_parity.py [envs] [periods_per_launch]"""
import json, sys
from collections import Counter, defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16
e = Engine(DqlConfig(dtype=F32, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1, fold_per_step=1), n, seed=42)
e.set_option("periods_per_launch", P)
e.train_steps(10 * P, 1.0); e.sync()
L = 6
e.episode_log_enable(L * P)
e.train_steps(L * P, 1.0); e.sync()
_, hw = e.episode_log_read()
hw = hw.astype(np.int64)[::P]
out = []
for j in range(L):
    h = hw[j]
    slot = h & 15; simd = (h >> 4) & 3; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7; xcc = (h >> 32) & 15
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    by = defaultdict(list)
    for w, (k, s) in enumerate(zip(key.tolist(), slot.tolist())):
        by[k].append((w, s))
    pairs = [v for v in by.values() if len(v) == 2]
    same_parity = sum(1 for v in pairs if (v[0][1] & 1) == (v[1][1] & 1))
    older_slot = Counter(min(v)[1] for v in pairs)     # slot of the wave with the smaller grid index (first dispatched)
    out.append({"simds_with_two_env_waves": len(pairs), "simds_with_other_counts": len(by) - len(pairs), "pairs_with_equal_slot_parity": same_parity,
                "slot_pairs": dict(Counter(tuple(sorted(s for _, s in v)) for v in pairs).most_common(6)).__repr__(), "slot_of_the_first_dispatched_wave": dict(older_slot)})
print(json.dumps({"envs": n, "periods_per_launch": P, "launches": out}))
e.close()
