"""Classify the VALU instructions of a kernel's loops by the issue cost measured in tools/micro/pk_variants.hip.

usage: isa_classify.py dql.s <mangled kernel name>
For every backward branch (a loop) in the kernel: instruction counts by class -- plain VGPR/literal operands, with an SGPR
source operand, packed (v_pk_*), transcendental (rcp/sqrt/rsq/...), 64-bit / f64, DPP, and non-VALU (SALU, memory, waitcnt).
"""
import re
import sys
from collections import Counter

src, kern = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
label_at = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label_at[m.group(1)] = i

TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")


def klass(l):
    t = l.strip().split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        return None
    op = t.split()[0]
    if not op.startswith("v_"):
        return "salu" if op.startswith("s_") else "mem"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(TRANS):
        return "valu_trans"
    if "_f64" in op or "_b64" in op or "_u64" in op or "_i64" in op:
        return "valu_64"
    if "dpp" in t or "row_" in t or "quad_perm" in t:
        return "valu_dpp"
    ops = t[len(op):]
    # an SGPR *source*: s<N>, s[a:b], vcc, exec, m0 anywhere after the destination
    parts = [p.strip() for p in ops.split(",")]
    srcs = parts[1:] if not op.startswith("v_cmp") else parts
    if op.startswith(("v_cndmask", "v_addc", "v_subb", "v_div_fmas")):
        return "valu_mask"
    if op.startswith("v_cmp"):
        return "valu_cmp" + ("_sgpr" if any(re.match(r"^-?\|?s\d|^-?\|?s\[", p) for p in parts[1:]) else "")
    if any(re.match(r"^-?\|?(s\d|s\[|vcc|exec|m0|ttmp)", p) for p in srcs):
        return "valu_sgpr"
    if any(re.match(r"^-?\|?(0x|[0-9.\-]+$)", p) for p in srcs):
        return "valu_const"
    return "valu_vgpr"


loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in label_at and label_at[tgt] < i:
            loops.append((label_at[tgt], i, tgt))
tot = Counter(k for k in map(klass, body) if k)
print("kernel", kern, dict(tot))
for a, b, t in sorted(loops, key=lambda x: x[0] - x[1])[:6]:
    c = Counter(k for k in map(klass, body[a:b + 1]) if k)
    n = sum(c.values())
    print(f"loop {t} lines {a}..{b}: {n} instr", dict(sorted(c.items())))
