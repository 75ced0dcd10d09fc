#!/usr/bin/env python3
"""G14 — the reference's own Gazebo flight records, as data.  Build container only (reads /root/reference, writes tests/golden/).

`pkg/trainer.py:247-303` opens a fresh `SummaryWriter` per finished episode, so every run directory under the reference's `assets/` holds one
TensorBoard event file per episode (`<run>/logs/events.out.tfevents.<unix time>.<host>.<pid>.<n>`): five scalars (Success Rate, Cumulative
Reward, Exploration Rate, Learning Rate, Mean reward) and the termination text (`CheckResult` value, `pkg/mdp.py:68-77`), all at step =
`_curriculum_episode_count`.  They are the only thing the reference holds about flights in the REAL simulator (Gazebo 11 + ODE + RotorS).

The files are TFRecord streams of `Event` protobufs; neither tensorboard nor protobuf is needed to read them — this script walks the wire
format by hand (varints, length-delimited fields) and checks every record's masked CRC-32C.  Output: `g14_gazebo_episodes.npz`, one row per
episode and run: episode number, wall time, the five scalars, termination code, and the episode length recovered as round(cumulative / mean).
Nothing of the reference's source text is copied: fixtures are numbers.

    python tests/golden/make_g14.py
"""
from __future__ import annotations

import struct
import sys
from pathlib import Path

import numpy as np

REF_ASSETS = Path("/root/reference/assets")
OUT = Path(__file__).resolve().parent / "g14_gazebo_episodes.npz"

# CheckResult values (pkg/mdp.py:68-77) -> the codes of include/dql.h (declaration order), -1 = unknown text
CODES = {
    "SUCCESS: Touched platform": 0,                                      # DQL_TERMINAL_CONTACT
    "SUCCESS: Goal state reached": 1,                                    # DQL_TERMINAL_SUCCESS
    "FAILURE: Drone moved too far from platform in x direction": 2,      # DQL_TERMINAL_FLYZONE_X
    "FAILURE: Drone moved too far from platform in y direction": 3,
    "FAILURE: Drone moved too far from platform in z direction": 4,
    "FAILURE: Reached minimum altitude": 5,
    "FAILURE: Maximum episode duration": 6,
}
TAGS = ("Episode/Success Rate", "Episode/Cumulative Reward", "Episode/Exploration Rate", "Episode/Learning Rate", "Episode/Mean reward")


def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _crc_table()


def masked_crc32c(b: bytes) -> int:
    c = 0xFFFFFFFF
    for x in b:
        c = _T[(c ^ x) & 0xFF] ^ (c >> 8)
    c ^= 0xFFFFFFFF
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def records(buf: bytes):
    """TFRecord framing: u64 length, u32 masked crc of the length, payload, u32 masked crc of the payload"""
    i = 0
    while i + 12 <= len(buf):
        (n,) = struct.unpack_from("<Q", buf, i)
        (c1,) = struct.unpack_from("<I", buf, i + 8)
        if masked_crc32c(buf[i:i + 8]) != c1:
            raise ValueError("length CRC mismatch")
        data = buf[i + 12:i + 12 + n]
        (c2,) = struct.unpack_from("<I", buf, i + 12 + n)
        if masked_crc32c(data) != c2:
            raise ValueError("payload CRC mismatch")
        yield data
        i += 16 + n


def fields(b: bytes):
    """protobuf wire format: (field number, wire type, value) — value is an int (varint), bytes (length-delimited) or raw 4 / 8 bytes"""
    i = 0
    while i < len(b):
        key, sh = 0, 0
        while True:
            x = b[i]; i += 1
            key |= (x & 0x7F) << sh; sh += 7
            if x < 0x80:
                break
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, sh = 0, 0
            while True:
                x = b[i]; i += 1
                v |= (x & 0x7F) << sh; sh += 7
                if x < 0x80:
                    break
            yield fno, wt, v
        elif wt == 1:
            yield fno, wt, b[i:i + 8]; i += 8
        elif wt == 2:
            n, sh = 0, 0
            while True:
                x = b[i]; i += 1
                n |= (x & 0x7F) << sh; sh += 7
                if x < 0x80:
                    break
            yield fno, wt, b[i:i + n]; i += n
        elif wt == 5:
            yield fno, wt, b[i:i + 4]; i += 4
        else:
            raise ValueError(f"wire type {wt}")


def parse_event_file(path: Path):
    """-> dict(step, wall_time, scalars {tag: float}, text)   Event: 1 wall_time (double), 2 step (int64), 5 summary; Summary: 1 value (repeated);
    Value: 1 tag, 2 simple_value (float), 8 tensor; TensorProto: 8 string_val (repeated bytes)"""
    out = {"step": None, "wall_time": None, "scalars": {}, "text": None}
    for rec in records(path.read_bytes()):
        step, wall, summary = 0, None, None
        for fno, wt, v in fields(rec):
            if fno == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif fno == 2 and wt == 0:
                step = v
            elif fno == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for fno, wt, val in fields(summary):
            if fno != 1 or wt != 2:
                continue
            tag, simple, tensor = None, None, None
            for f2, w2, v2 in fields(val):
                if f2 == 1 and w2 == 2:
                    tag = v2.decode()
                elif f2 == 2 and w2 == 5:
                    (simple,) = struct.unpack("<f", v2)
                elif f2 == 8 and w2 == 2:
                    tensor = v2
            if simple is not None:
                out["scalars"][tag] = simple
            elif tensor is not None:
                for f3, w3, v3 in fields(tensor):
                    if f3 == 8 and w3 == 2:
                        out["text"] = v3.decode()
            out["step"] = step
            if out["wall_time"] is None:
                out["wall_time"] = wall
    return out


def run_dirs():
    for logs in sorted(REF_ASSETS.glob("**/logs")):
        if logs.is_dir():
            yield str(logs.parent.relative_to(REF_ASSETS)), logs


def main():
    cols = {k: [] for k in ("run", "file_index", "episode", "wall_time", "success_rate", "cumulative_reward", "exploration_rate", "learning_rate", "mean_reward", "code")}
    runs, unknown = [], {}
    for name, logs in run_dirs():
        files = sorted(logs.glob("events.out.tfevents.*"), key=lambda p: (int(p.name.split(".")[3]), int(p.name.rsplit(".", 1)[1])))
        rid = len(runs); runs.append(name); n_ok = 0
        for p in files:
            try:
                ev = parse_event_file(p)
            except (ValueError, IndexError, struct.error) as e:
                print(f"skip {p.name}: {e}", file=sys.stderr)
                continue
            if ev["step"] is None or len(ev["scalars"]) < 5:
                continue
            code = CODES.get(ev["text"], -1)
            if code < 0:
                unknown[ev["text"]] = unknown.get(ev["text"], 0) + 1
            cols["run"].append(rid); cols["file_index"].append(int(p.name.rsplit(".", 1)[1])); cols["episode"].append(ev["step"]); cols["wall_time"].append(ev["wall_time"])
            for k, tag in zip(("success_rate", "cumulative_reward", "exploration_rate", "learning_rate", "mean_reward"), TAGS):
                cols[k].append(ev["scalars"][tag])
            cols["code"].append(code); n_ok += 1
        print(f"{name}: {n_ok} episodes of {len(files)} files")
    if unknown:
        print("unknown termination texts:", unknown, file=sys.stderr)
    arr = {"run": np.array(cols["run"], np.int16), "file_index": np.array(cols["file_index"], np.int32), "episode": np.array(cols["episode"], np.int32),
           "wall_time": np.array(cols["wall_time"], np.float64), "code": np.array(cols["code"], np.int8)}
    for k in ("success_rate", "cumulative_reward", "exploration_rate", "learning_rate", "mean_reward"):
        arr[k] = np.array(cols[k], np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        steps = np.rint(arr["cumulative_reward"].astype(np.float64) / arr["mean_reward"].astype(np.float64))
    # "Mean reward" = cumulative / step_count, both written by check() BEFORE reward() adds the terminal step's reward (pkg/mdp.py:435-438,
    # pkg/landing_simulation_env.py:274-275): the logged cumulative reward is the sum over steps 1 .. n-1, and an episode that ends at its
    # first step logs 0 / 1 = 0 for both
    first_step = (arr["cumulative_reward"] == 0) & (arr["mean_reward"] == 0)
    arr["steps"] = np.where(first_step, 1, np.where(np.isfinite(steps), steps, -1)).astype(np.int32)
    arr["runs"] = np.array(runs)
    # a run directory that is a byte-for-byte copy of another one's records (assets/x_direction/<run> == assets/<run>) is listed, not stored twice
    dup = np.full(len(runs), -1, np.int16)
    keep = np.ones(len(arr["run"]), bool)
    for b in range(len(runs)):
        for a in range(b):
            ma, mb = arr["run"] == a, arr["run"] == b
            if dup[a] < 0 and ma.sum() == mb.sum() and all(np.array_equal(arr[k][ma], arr[k][mb]) for k in arr if k not in ("run", "runs")):
                dup[b] = a; keep &= ~mb
                break
    for k in list(arr):
        if k != "runs":
            arr[k] = arr[k][keep]
    arr["run_duplicate_of"] = dup
    np.savez_compressed(OUT, **arr)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes, {len(arr['run'])} episodes, {len(runs)} runs)")


if __name__ == "__main__":
    main()
