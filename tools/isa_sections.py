#!/usr/bin/env python3
"""Static instruction counts per marked section of the tick loop (build with -DDQL_MARK), split by the issue-cost class measured in
tools/micro/pk_variants.hip + valu_forms.hip (ns per wave64 instruction per SIMD with >= 2 waves resident):
  fast   v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub f32, v_mov, v_and with VGPR / literal sources      ~1.0-1.25
  slow   every other 32-bit VALU opcode (compare, select, max/min/med3, conversions, integer), and ANY opcode with an
         SGPR source operand                                                                                          ~1.7-1.9
  pk     v_pk_*_f32 (two operations per lane)                                                                           ~1.9-2.1
  trans  v_rcp / v_sqrt / v_rsq                                                                                         ~3.5
usage: isa_sections.py [mangled kernel name] [extra hipcc flags]
"""
import re, subprocess, sys
from collections import Counter, defaultdict
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "dql_multirotor_landing_amd" / "csrc" / "dql_hip.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DDQL_MARK", *sys.argv[2:], "--cuda-device-only", "-S", str(src), "-o", "/tmp/dql_mark.s"], check=True, capture_output=True)
s = open("/tmp/dql_mark.s").read()
name = sys.argv[1] if len(sys.argv) > 1 else "_Z6k_stepIfLi256ELi0EEv8StepArgsIT_E"
a = s.index(name + ":"); b = s.index(".Lfunc_end", a)
FAST = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32")
TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
COST = {"fast": 1.15, "slow": 1.8, "sgpr": 1.85, "pk": 2.0, "trans": 3.5, "f64": 3.6}


def klass(t):
    op = t.split()[0]
    if not op.startswith("v_"):
        return "salu" if op.startswith("s_") else "mem"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(TRANS):
        return "trans"
    if "_f64" in op:
        return "f64"
    parts = [p.strip() for p in t[len(op):].split(",")]
    if any(re.match(r"^-?\|?(s\d|s\[|vcc|exec|m0)", p) for p in parts[1:]) and not op.startswith("v_cmp"):
        return "sgpr"
    return "fast" if op.startswith(FAST) else "slow"


cur = "prologue"; occ = Counter({"prologue": 1}); cls = defaultdict(Counter)
for l in s[a:b].split("\n"):
    m = re.search(r"; SECTION (\w+)", l)
    if m: cur = m.group(1); occ[cur] += 1; continue
    t = l.strip().split(";")[0].strip()
    if l.startswith("\t") and t and not t.startswith("."):
        cls[cur][klass(t)] += 1
print(f"{'section':18s} {'copies':>6s} " + " ".join(f"{k:>6s}" for k in ("fast", "slow", "sgpr", "pk", "trans", "f64", "salu", "mem")) + "   VALU/copy  est ns/copy (>= 2 waves per SIMD)")
for k in cls:
    n = occ[k]; c = cls[k]
    valu = sum(c[x] for x in COST)
    ns = sum(c[x] * COST[x] for x in COST)
    print(f"{k:18s} {n:6d} " + " ".join(f"{c[x]:6d}" for x in ("fast", "slow", "sgpr", "pk", "trans", "f64", "salu", "mem")) + f"   {valu / n:9.1f}  {ns / n:8.1f}")
