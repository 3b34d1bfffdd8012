#!/usr/bin/env python
"""Registers, spills and scratch of every kernel in a hipcc -save-temps .s file (developer tool).
    python tools/kernel_regs.py path/to/file-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    agpr, name, priv, sgpr, vgpr, spill = m.groups()
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    if flt in name:
        rows.append((name, int(vgpr), int(agpr), int(spill), int(priv)))
print("%-70s %5s %5s %6s %8s" % ("kernel", "vgpr", "agpr", "spill", "scratchB"))
for r in rows:
    print("%-70s %5d %5d %6d %8d" % r)
