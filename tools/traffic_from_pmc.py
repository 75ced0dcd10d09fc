#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs of tools/prof_run.py) -> profiles/r1_traffic.json.

    python tools/traffic_from_pmc.py gpurun_out/pmc_{fetch,write}_{4096,1048576} > profiles/r1_traffic.json
Directory names end in _<counter>_<envs>.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section):
counters in KiB, FETCH_SIZE reports half of a wide coalesced read stream -> doubled; WRITE_SIZE as is."""
import csv, glob, json, sys
from collections import defaultdict
cfg = defaultdict(dict)
for d in sys.argv[1:]:
    kind, envs = d.rstrip("/").split("_")[-2:]
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_step" in r["Kernel_Name"] and r["Counter_Name"] == ("FETCH_SIZE" if kind == "fetch" else "WRITE_SIZE")]
    vals = vals[5:]  # warm launches
    cfg[envs]["FETCH_SIZE_KiB_raw_per_launch" if kind == "fetch" else "WRITE_SIZE_KiB_raw_per_launch"] = sum(vals) / len(vals)
for envs, c in cfg.items():
    c["read_bytes_per_launch"] = 2 * 1024 * c["FETCH_SIZE_KiB_raw_per_launch"]
    c["write_bytes_per_launch"] = 1024 * c["WRITE_SIZE_KiB_raw_per_launch"]
    c["hbm_bytes_per_launch"] = c["read_bytes_per_launch"] + c["write_bytes_per_launch"]
    c["envs"] = int(envs)
    c["hbm_bytes_per_env"] = c["hbm_bytes_per_launch"] / int(envs)
print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), tools/prof_run.py, kernel k_step, averages over launches 6..N",
                  "units": "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read stream); WRITE_SIZE taken as is",
                  "configs": dict(cfg)}, indent=1))
