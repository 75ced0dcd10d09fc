#!/usr/bin/env python3
"""Where the dispatcher puts the env waves of one launch: per-wave HW_ID / XCC_ID stamps (diagnostic build -DDQL_WAVE_CLOCK=8,
loaded through DQL_LIB_PATH) -> waves per SIMD, workgroups per CU.

    tools/ab_build.sh clock8 -DDQL_WAVE_CLOCK=8
    DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip_clock8.so python tools/exp_placement.py 65536,131072
gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0] in its own register.
"""
import json, sys
from collections import Counter
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from dql_multirotor_landing_amd.config import DqlConfig, F32
from dql_multirotor_landing_amd.engine import Engine
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["65536", "131072"])]:
    e = Engine(DqlConfig(dtype=F32), n, seed=42)
    if len(sys.argv) > 2:
        e.set_option("block", int(sys.argv[2]))
    e.train_steps(50, 1.0); e.sync()
    K = 20
    e.episode_log_enable(K)
    e.timer_start(); e.train_steps(K, 1.0); ms = e.timer_stop()
    t0, hw = e.episode_log_read()
    hw = hw.astype(np.int64)
    rows = []
    for j in range(K):
        h = hw[j]
        simd = (h >> 4) & 3; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7; xcc = (h >> 32) & 15
        cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        per_simd = Counter((cu_key * 4 + simd).tolist())
        per_cu = Counter(cu_key.tolist())
        rows.append({"cus_used": len(per_cu), "simds_used": len(per_simd), "max_waves_per_simd": max(per_simd.values()),
                     "waves_per_simd_hist": dict(sorted(Counter(per_simd.values()).items())), "waves_per_cu_hist": dict(sorted(Counter(per_cu.values()).items())),
                     "xcc_hist": dict(sorted(Counter(xcc.tolist()).items()))})
    print(json.dumps({"envs": n, "waves": int(hw.shape[1]), "us_per_step": ms * 1e3 / K, "launch0": rows[0], "launch_last": rows[-1],
                      "max_waves_per_simd_over_launches": [r["max_waves_per_simd"] for r in rows]}), flush=True)
    e.close()
