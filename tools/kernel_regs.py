#!/usr/bin/env python3
"""Register / LDS / scratch use per kernel from a hipcc -S listing (metadata section).
usage: kernel_regs.py file.s [substring]"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = f.get("name", "?")
    if want in name:
        print("%-110s vgpr %3s agpr %s spill v%s s%s lds %6s scratch %s" % (
            name[:110], f.get("vgpr_count"), blk.split()[0], f.get("vgpr_spill_count"), f.get("sgpr_spill_count"),
            f.get("group_segment_fixed_size"), f.get("private_segment_fixed_size")))
