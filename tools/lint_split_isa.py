#!/usr/bin/env python3
"""Build-time lint of cb_split.hip's contraction kernels.  Their fragment reads are inline-asm ds_read_b128 whose
results the compiler takes for valid at once, while they only arrive behind the next `s_waitcnt lgkmcnt(0)`.  Any
instruction the compiler places in between that touches such a register -- a spill to scratch, a copy at a
branch join -- would move garbage (round 3: wrong tiles, only under load).  This script compiles the file to
assembly and checks, for every cbs_conv_kernel instance, that between a ds_read_b128 and the following
s_waitcnt lgkmcnt(0) no other instruction names a register a pending read writes.
usage: lint_split_isa.py [path/to/cb_split.hip]   (exit code 1 on a finding)"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def lint(asm):
    findings, kernels = [], 0
    cur, pending, name, in_asm = None, set(), None, False
    for ln, line in enumerate(asm.splitlines(), 1):
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"^(_ZN3cbs15cbs_conv_kernel\S*):", line)
        if m:
            name, pending, cur = m.group(1), set(), True
            kernels += 1
            continue
        if not cur:
            continue
        ins = line.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        if ins.startswith("s_endpgm"):
            cur = None
            continue
        op = ins.split()[0]
        if op == "ds_read_b128" and in_asm:      # (the compiler's own LDS reads are waited for by the compiler)
            dst = ins.split(None, 1)[1].split(",")[0]
            touched = regs(ins.split(",", 1)[1]) & pending
            if touched:
                findings.append((name, ln, ins, sorted(touched)))
            pending |= regs(dst)
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in ins:
            pending = set()
            continue
        if pending and (regs(ins) & pending):
            findings.append((name, ln, ins, sorted(regs(ins) & pending)))
    return kernels, findings


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "cbinfer_amd", "csrc", "cb_split.hip")
    flags = os.environ.get("LINT_FLAGS", "").split()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S",
                               "-I", os.path.dirname(src), src, "-o", out] + flags, stderr=subprocess.DEVNULL)
        kernels, findings = lint(open(out).read())
    print("%d cbs_conv_kernel instance(s), %d finding(s)" % (kernels, len(findings)))
    for f in findings[:20]:
        print("  %s line %d: %s  (pending %s)" % f)
    return 1 if findings or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
