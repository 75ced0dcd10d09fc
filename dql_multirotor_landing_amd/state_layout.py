"""What the per-env state fields MEAN per dtype, and the one mapping between the dtypes' layouts.

All 64 real fields mean the same in float32 and float64 contexts except the two PIDs' Butterworth filter fields
(`vz_x1, vz_x2, vz_y1, vz_y2, vz_y3` and `yw_*`):

* float64 keeps the reference's histories (pkg/filters.py:98-109): x1, x2 = the last two inputs, y1 .. y3 = the last three outputs.
* float32 keeps the three states t1, t2, t3 of the same recurrence in transposed direct form in (x1, x2, y1); y2 = y3 = 0
  (csrc/dql_device.hpp `butterworth`):  t1 = 2b x1 + b x2 - a2 y2 - a3 y3,  t2 = b x1 - a2 y1 - a3 y2,  t3 = -a3 y1.

`STATE_LAYOUT_VERSION` is written into every checkpoint next to the dtype (trainer.py); a float64 env state can be mapped onto the
float32 layout (`to_f32_filter_state`), the reverse is not unique (three numbers do not determine five) and is refused.
"""
from __future__ import annotations

import numpy as np

STATE_LAYOUT_VERSION = 2   # 1: rounds 1-3 (histories in both dtypes); 2: round 4+ (float32: transposed states)
_FILTERS = ("vz_", "yw_")


def butterworth_coefficients(bw_c: float = 1.0):
    denom = 1 + bw_c * bw_c + 1.414 * bw_c                     # pkg/filters.py:94
    return 1.0 / denom, (-2 * bw_c * bw_c + 2) / denom, (bw_c * bw_c - 1.414 * bw_c + 1) / denom   # b, a2, a3


def to_f32_filter_state(reals, names, bw_c: float = 1.0):
    """float64-layout env state [n_fields][n] -> the float32 layout's (a copy; only the ten filter fields change)"""
    r = np.array(reals, dtype=np.float64, copy=True)
    b, a2, a3 = butterworth_coefficients(bw_c)
    for pfx in _FILTERS:
        x1, x2, y1, y2, y3 = (np.asarray(reals[names.index(pfx + k)], dtype=np.float64) for k in ("x1", "x2", "y1", "y2", "y3"))
        r[names.index(pfx + "x1")] = 2 * b * x1 + b * x2 - a2 * y2 - a3 * y3
        r[names.index(pfx + "x2")] = b * x1 - a2 * y1 - a3 * y2
        r[names.index(pfx + "y1")] = -a3 * y1
        r[names.index(pfx + "y2")] = 0.0
        r[names.index(pfx + "y3")] = 0.0
    return r


def convert_env_state(reals, names, src_dtype: int, dst_dtype: int, src_version: int = STATE_LAYOUT_VERSION, bw_c: float = 1.0):
    """env state saved by a context of `src_dtype` (0 = float32, 1 = float64) at layout `src_version` -> what a context of
    `dst_dtype` expects, or None when no mapping exists (the caller then resumes from the tables alone)."""
    src_hist = src_dtype == 1 or src_version < 2     # the saved filter fields are the reference's histories
    dst_hist = dst_dtype == 1
    if src_hist == dst_hist:
        return np.asarray(reals, dtype=np.float64)
    if src_hist and not dst_hist:
        return to_f32_filter_state(reals, names, bw_c)
    return None
