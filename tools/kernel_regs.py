#!/usr/bin/env python3
"""Registers / spills per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks:  kernel_regs.py <stderr file> [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
SCR = r'ScratchSize \[bytes/lane\]'
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    if flt not in name:
        continue
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    print(f"{dem[:80]:80s} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):3d} SGPR {g('TotalSGPRs'):3d} scratch {g(SCR):4d} "
          f"spill v {g('VGPRs Spill'):3d} s {g('SGPRs Spill'):3d}")
