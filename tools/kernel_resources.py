#!/usr/bin/env python3
"""Prints VGPR/SGPR/LDS/occupancy per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` logs."""
import re
import sys

OCC = r"Occupancy \[waves/SIMD\]"
LDS = r"LDS Size \[bytes/block\]"


def get(b, k):
    m = re.search(k + r": (\d+)", b)
    return m.group(1) if m else "?"


for f in sys.argv[1:]:
    txt = open(f).read()
    for b in txt.split("Function Name: ")[1:]:
        name = b.split(" ")[0]
        short = re.sub(r"_ZN4xpic12_GLOBAL__N_1\d+", "", name)[:44]
        print("%-46s vgpr=%4s sgpr=%4s occ=%2s lds=%6s spill=%s" % (
            short, get(b, "VGPRs"), get(b, "TotalSGPRs"), get(b, OCC), get(b, LDS), get(b, "VGPRs Spill")))
