#!/usr/bin/env python3
"""Static instruction counts per marked section of the tick loop (build with -DDQL_MARK)."""
import re, subprocess, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "dql_multirotor_landing_amd" / "csrc" / "dql_hip.hip"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DDQL_MARK", *sys.argv[2:], "--cuda-device-only", "-S", str(src), "-o", "/tmp/dql_mark.s"], check=True, capture_output=True)
s = open("/tmp/dql_mark.s").read()
name = sys.argv[1] if len(sys.argv) > 1 else "_Z6k_stepIfLi64ELb1EEv8StepArgsIT_E"
a = s.index(name + ":"); b = s.index(".Lfunc_end", a)
cur = "prologue"; counts = Counter(); valu = Counter()
for l in s[a:b].split("\n"):
    m = re.search(r"; SECTION (\w+)", l)
    if m: cur = m.group(1); continue
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith(".") and not t.startswith(";"):
        counts[cur] += 1
        if t.startswith("v_"): valu[cur] += 1
for k in counts: print(f"{k:18s} total {counts[k]:5d}  valu {valu[k]:5d}")
