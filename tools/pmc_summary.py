#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/prof_run.py -> profiles/<round>_pmc_sq_summary.json and profiles/<round>_traffic.json (round = $DQL_ROUND,
default r3), stamped with the hash of the kernel sources the passes ran (bench.py refuses a traffic figure whose stamp is not the library's).

    python tools/pmc_summary.py gpurun_out/<dir> ...        # dirs named pmc_<sq|fetch|write>_<envs>_p<P>[_cfg4|_2axis]
Every pass is its own run (`rocprofv3 --kernel-trace --pmc ... -- python3 tools/prof_run.py N S 0 P`); per-launch averages of
k_step over launches 6..N (the first launches hold the reset period).  Units and the gfx950 correction follow
MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950 reports half of a wide
coalesced read stream), WRITE_SIZE as is."""
import csv, glob, json, os, re, sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import lib_source_sha16  # noqa: E402
ROUND = os.environ.get("DQL_ROUND", "r3")
sq, tr = {}, defaultdict(dict)
for d in sys.argv[1:]:
    m = re.search(r"pmc_(sq|fetch|write)_(\d+)_p(\d+)(_cfg4|_2axis)?$", d.rstrip("/"))
    kind, envs, P, flav = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4) or ""
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=lambda x: Path(x).stat().st_mtime)  # the newest pass in the directory
    per = defaultdict(list); kname = None
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"])); kname = r["Kernel_Name"]
    key = f"{envs}_p{P}{flav}"
    if kind == "sq":
        c = {k: sum(v[5:]) / len(v[5:]) for k, v in per.items()}
        c.update(periods_per_launch=P, launches_averaged=len(next(iter(per.values()))) - 5, kernel=kname,
                 SQ_INSTS_VALU_per_env_wave_per_period=c["SQ_INSTS_VALU"] / ((envs + 63) // 64) / P,
                 note="per launch, summed over all waves incl. the table-writer workgroups")
        sq[key] = c
    else:
        name = "FETCH_SIZE" if kind == "fetch" else "WRITE_SIZE"
        v = per[name][5:]
        tr[key][f"{name}_KiB_raw_per_launch"] = sum(v) / len(v)
        tr[key].update(envs=envs, periods_per_launch=P, kernel=kname)
for key, c in tr.items():
    if "FETCH_SIZE_KiB_raw_per_launch" not in c or "WRITE_SIZE_KiB_raw_per_launch" not in c:
        continue
    n = c["envs"] * c["periods_per_launch"]
    c["read_bytes_per_launch"] = 2 * 1024 * c["FETCH_SIZE_KiB_raw_per_launch"]
    c["write_bytes_per_launch"] = 1024 * c["WRITE_SIZE_KiB_raw_per_launch"]
    c["hbm_bytes_per_launch"] = c["read_bytes_per_launch"] + c["write_bytes_per_launch"]
    c["read_bytes_per_env_step"] = c["read_bytes_per_launch"] / n
    c["write_bytes_per_env_step"] = c["write_bytes_per_launch"] / n
    c["hbm_bytes_per_env_step"] = c["hbm_bytes_per_launch"] / n
    algo = 400.0 if key.endswith("_2axis") else (328.0 if key.endswith("_cfg4") else 320.0)
    c["algorithmic_bytes_per_env_step"] = algo
    c["ratio_to_algorithmic"] = c["hbm_bytes_per_env_step"] / algo
if sq:
    (ROOT / "profiles" / f"{ROUND}_pmc_sq_summary.json").write_text(json.dumps({"source_sha16": lib_source_sha16(), "configs": dict(sorted(sq.items()))}, indent=1))
if tr:
    (ROOT / "profiles" / f"{ROUND}_traffic.json").write_text(json.dumps({
        "source_sha16": lib_source_sha16(),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), tools/prof_run.py N S 0 P, kernel k_step, averages over launches 6..N",
        "units": "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); WRITE_SIZE taken as is; per env-step = per launch / (envs x periods per launch)",
        "configs": dict(sorted(tr.items()))}, indent=1))
print("sq:", sorted(sq), "traffic:", sorted(tr))
