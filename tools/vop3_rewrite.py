#!/usr/bin/env python3
"""Experiment (lost): re-encode VOP1 / VOP2 `_e32` instructions of chosen kernels as `_e64` in the device assembly.
    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S dql_hip.hip -o dev.s; python tools/vop3_rewrite.py dev.s dev2.s k_stepIfLi64ELi2E,k_stepIfLi256ELi2E
    clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c dev2.s; lld -flavor gnu -m elf64_amdgpu --no-undefined -shared; clang-offload-bundler ...;
    hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang dev2.hipfb -c dql_hip.hip; hipcc -shared"""
import re, sys
src, dst = sys.argv[1], sys.argv[2]
targets = sys.argv[3].split(",")  # substrings of kernel symbol names to rewrite
INLINE = re.compile(r"^-?(0\.5|1\.0|2\.0|4\.0|0|[1-9]|[1-5][0-9]|6[0-4]|-1[0-6]|-[1-9])$")
SKIP_OPS = ("v_fmaak", "v_fmamk", "v_madak", "v_madmk", "v_readfirstlane", "v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_accvgpr", "v_nop", "v_swap")
def is_lit(op):
    op = op.strip()
    if re.match(r"^(v|s|a)\d+$|^(v|s|a)\[\d+:\d+\]$|^vcc(_lo|_hi)?$|^exec|^\|?-?\|?v\d+\|?$|^-v\d+$|^-s\d+$|^m0$|^scc$|^src_", op): return False
    if INLINE.match(op): return False
    return True
def is_sgpr(op):
    return bool(re.match(r"^-?s\d+$|^s\[\d+:\d+\]$", op.strip()))
n_conv = n_skip = 0
out = []
cur = None
for line in open(src):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1)
    if line.startswith(".Lfunc_end"): cur_end = True
    s = line.strip()
    if cur and any(t in cur for t in targets) and s.startswith("v_") and "_e32" in s.split()[0]:
        op = s.split()[0]
        body = s.split(";")[0]
        operands = body[len(op):].split(",")
        base = op[:-4]
        ok = not any(base.startswith(k) for k in SKIP_OPS) and "dpp" not in body and "sdwa" not in body
        if ok and any(is_lit(o) for o in operands[1:] if o.strip()): ok = False
        uses_vcc = any(o.strip().startswith("vcc") for o in operands)
        n_s = len({o.strip().lstrip("-") for o in operands[1:] if is_sgpr(o)})
        if ok and (n_s + (1 if uses_vcc and (base.startswith("v_cndmask") or "addc" in base or "subb" in base) else 0)) > 1: ok = False
        if ok:
            line = line.replace(op, base + "_e64", 1); n_conv += 1
        else:
            n_skip += 1
    if line.startswith(".Lfunc_end"): cur = None
    out.append(line)
open(dst, "w").write("".join(out))
print("converted", n_conv, "left", n_skip)
