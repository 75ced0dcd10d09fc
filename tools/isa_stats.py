#!/usr/bin/env python3
"""Instruction statistics of a kernel in hipcc's gfx950 assembly: totals, histogram, loop spans.
usage: tools/isa_stats.py [kernel-substring]   (compiles csrc/dql_hip.hip to /tmp/dql.s first)"""
import re, subprocess, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "dql_multirotor_landing_amd" / "csrc" / "dql_hip.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", *sys.argv[2:], "--cuda-device-only", "-S", str(src), "-o", "/tmp/dql.s"], check=True, capture_output=True)
s = open("/tmp/dql.s").read()
name = sys.argv[1] if len(sys.argv) > 1 else "_Z6k_stepIfLi64ELb1EEv8StepArgsIT_E"
a = s.index(name + ":"); b = s.index(".Lfunc_end", a)
lines = s[a:b].split("\n")
def is_ins(l): return l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";")
ins = [l.strip() for l in lines if is_ins(l)]
print("total instructions", len(ins))
print(Counter(i.split()[0] for i in ins).most_common(30))
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            body = [x.strip() for x in lines[labels[t]:i] if is_ins(x)]
            loops.append((len(body), t, body))
loops.sort(key=lambda x: -x[0])
for n, t, body in loops[:4]:
    c = Counter(x.split()[0] for x in body)
    print(f"loop {t}: {n} instr; readlane {c['v_readlane_b32']} writelane {c['v_writelane_b32']} v_mov {c['v_mov_b32_e32']} "
          f"saveexec {sum(v for k, v in c.items() if 'saveexec' in k)} sqrt {c['v_sqrt_f32_e32']} rcp {c['v_rcp_f32_e32']} div_scale {c['v_div_scale_f32']} s_ {sum(v for k, v in c.items() if k.startswith('s_'))}")
    print("   ", c.most_common(25))
