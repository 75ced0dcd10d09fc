#!/usr/bin/env python3
"""PMC passes of tools/campaign_r5o.sh -> per env wave and agent period figures + shares of the wave cycles (the layout of profiles/r5_pmc_wave_cycle_breakdown.json):
    python tools/pmc_breakdown.py gpurun_out/r5o/pmc_*   (131 072 envs = 2 048 env waves, 16 periods per launch)"""
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import lib_source_sha16  # noqa: E402
c = {}
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=lambda x: Path(x).stat().st_mtime))):
        if "k_step" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        c[k] = sum(v[5:]) / max(1, len(v[5:]))
W, P = 2048, 16
inst = {k: round(v / (W * P), 1) for k, v in c.items() if k.startswith("SQ_INSTS_")}
wc = c.get("SQ_WAVE_CYCLES")
share = {k: round(c[k] / wc, 4) for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                                          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_SALU") if wc and k in c}
print(json.dumps({"source_sha16": lib_source_sha16(), "per_env_wave_per_period": inst, "wave_quad_cycles_per_env_wave_per_period": round(wc / (W * P), 1) if wc else None,
                  "share_of_wave_cycles": share}, indent=1))
