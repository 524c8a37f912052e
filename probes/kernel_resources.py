#!/usr/bin/env python3
"""summarize `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (name filter as argv[2])"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    if flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"pkv::\(anonymous namespace\)::", "", dn)
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{dn[:100]:100s} sgpr {g('TotalSGPRs'):>3} vgpr {g('VGPRs'):>3} agpr {g('AGPRs'):>3} scratch {scr:>4} spill {g('VGPRs Spill'):>3} occ {occ}")
