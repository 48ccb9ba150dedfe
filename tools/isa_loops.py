#!/usr/bin/env python3
"""Instruction mix of a kernel per loop-nest depth (the compiler's own loop annotations in a `hipcc -S` file):
isa_loops.py <file.s> <kernel name fragment>.  Tells the static counts of the hot loops from those of the prologue."""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and frag in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))  # (not the first s_endpgm: early exits)
cur, stats = 0, {}
for l in L[start:end]:
    m = re.search(r"Loop Header: Depth=(\d+)", l) or re.search(r"in Loop: Header=\S+ Depth=(\d+)", l)
    if m:
        cur = int(m.group(1))
    elif re.match(r"^\.LBB\d+_\d+:\s*$", l):
        cur = 0  # a block outside every loop carries no annotation
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.split(";")[0].strip().endswith(":"):
        continue
    op = t.split()[0]
    key = ("readlane" if "v_readlane" in op else "writelane" if "v_writelane" in op else "waitcnt" if op == "s_waitcnt" else
           "salu" if op.startswith("s_") else "scratch" if op.startswith("scratch_") else
           "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "lds" if op.startswith("ds_") else
           "mfma" if "mfma" in op else "valu")
    d = stats.setdefault(cur, {})
    d[key] = d.get(key, 0) + 1
for k in sorted(stats):
    print("depth", k, dict(sorted(stats[k].items())))
