#!/usr/bin/env python3
"""rocprofv3 --pmc passes with the wave-cycle counters (tools/campaign_r4.sh) -> one JSON: per-launch averages of k_step over launches 6..N
and the ratios VERDICT r3 asked for (cycles per VALU instruction per wave, share of wave cycles spent waiting, transcendental share)."""
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import lib_source_sha16  # noqa: E402
out = {"source_sha16": lib_source_sha16(), "workload": "tools/prof_run.py 131072 320 0 16 cfg4 (k_step, averages over launches 6..N)", "counters": {}}
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=lambda x: Path(x).stat().st_mtime))):
        if "k_step" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"])); out["kernel"] = r["Kernel_Name"]
    for k, v in per.items():
        out["counters"][k] = sum(v[5:]) / max(1, len(v[5:]))
c = out["counters"]
g = lambda k: c.get(k)
if g("SQ_WAVE_CYCLES") and g("SQ_INSTS_VALU"):
    out["wave_cycles_per_valu_instruction"] = g("SQ_WAVE_CYCLES") / g("SQ_INSTS_VALU")
if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
    out["share_of_wave_cycles_waiting_for_any_instruction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    out["share_of_wave_cycles_with_a_valu_instruction_active"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
if g("SQ_INSTS_VALU_TRANS") and g("SQ_INSTS_VALU"):
    out["transcendental_share_of_valu_instructions"] = g("SQ_INSTS_VALU_TRANS") / g("SQ_INSTS_VALU")
print(json.dumps(out, indent=1))
