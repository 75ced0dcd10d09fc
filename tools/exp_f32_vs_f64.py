"""float32 HIP kernel vs float64 oracle from identical states, per field, over 1 period and over a 16-period launch, on the config
flavours the throughput figures run on.  Prints one JSON line per (case, periods): max error per field group —
the numbers tests/test_gpu_parity.py's tolerances are set from.  usage: python tools/exp_f32_vs_f64.py [n_envs]"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dql_multirotor_landing_amd.config import DqlConfig, F32, F64  # noqa: E402
from dql_multirotor_landing_amd.engine import Engine  # noqa: E402
from dql_multirotor_landing_amd.state_layout import to_f32_filter_state  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

CASES = {
    "default": {},
    "configs4": dict(per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1),
    "two_axis": dict(two_axis=1),
    "two_axis_configs4": dict(two_axis=1, per_env_platform=1, noise_pos_sd=0.25, noise_vel_sd=0.1),
}
GROUPS = {
    "pose": ["px", "py", "pz", "qw", "qx", "qy", "qz"], "vel": ["vx", "vy", "vz", "wx", "wy", "wz"], "rotor": ["om0", "om1", "om2", "om3"],
    "pid": ["vz_i", "yw_i", "vz_state", "yw_state"], "platform": ["mp_phase", "mp_x", "mp_u", "mp_y", "mp_v"],
    "obs_pv": ["obs_p_x", "obs_v_x", "obs_p_y", "obs_v_y"], "obs_a": ["obs_a_x", "obs_a_y", "kal_x_x", "kal_y_x"], "kal_P": ["kal_x_P", "kal_y_P"],
    "reward": ["reward"], "cum": ["cum_x", "cum_y"], "shaping": ["shp_x_p", "shp_x_v", "shp_x_a", "shp_y_p", "shp_y_v", "shp_y_a"], "sp": ["pitch_sp", "roll_sp"],
}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    for case, kw in CASES.items():
        for periods in (1, 16):
            o64 = Oracle(DqlConfig(dtype=F64, **kw), n, seed=9, n_threads=16)
            o64.train_steps(50, 1.0)
            reals, ints = o64.get_fields()
            e32 = Engine(DqlConfig(dtype=F32, **kw), n, seed=9)
            e32.train_steps(50, 1.0)
            names = e32.field_names()
            qa, qb, cnt = o64.qa.copy(), o64.qb.copy(), o64.count.copy()
            e32.set_tables(qa, qb, cnt); o64.set_tables(qa, qb, cnt)
            e32.set_fields(to_f32_filter_state(reals, names), ints)
            e32.set_option("periods_per_launch", periods); o64.set_option("periods_per_launch", periods)
            e32.train_steps(periods, 1.0); o64.train_steps(periods, 1.0)
            r32, i32 = e32.get_fields(); r64, i64 = o64.get_fields()
            inames = e32.field_names(True)
            same = np.ones(n, dtype=bool)
            for k in ("step_count", "code", "flags", "cur_check"):
                same &= i32[inames.index(k)] == i64[inames.index(k)]
            row = {"case": case, "periods": periods, "n": n, "envs_with_same_episode_state": float(same.mean()),
                   "idx_x_mismatch": float((i32[0] != i64[0])[same].mean()), "idx_y_mismatch": float((i32[1] != i64[1])[same].mean()),
                   "resets_in_window": float(((i64[inames.index("flags")] & 8) != 0).mean())}
            for g, fields in GROUPS.items():
                worst_abs, worst_rel = 0.0, 0.0
                for f in fields:
                    k = names.index(f)
                    d = np.abs(r32[k] - r64[k])[same]
                    if d.size == 0:
                        continue
                    worst_abs = max(worst_abs, float(d.max()))
                    worst_rel = max(worst_rel, float((d / np.maximum(np.abs(r64[k][same]), 1.0)).max()))
                row[g] = [worst_abs, worst_rel]
            print(json.dumps(row), flush=True)
            e32.close()


if __name__ == "__main__":
    main()
