#!/usr/bin/env python3
"""Instruction mix of a kernel in a hipcc -save-temps .s file: isa_count.py <file.s> <kernel name fragment> [from regex] [to regex]"""
import re
import sys
from collections import Counter

L = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, l in enumerate(L) if l.startswith("_Z") and frag in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
body = [l.strip() for l in L[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def cat(i):
    op = i.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if "_f64" in op and op.startswith("v_") and not op.startswith("v_cmp"):
        return "dp"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu32"
    return "other"


a = 0
b = len(body)
if len(sys.argv) > 3:
    a = next(i for i in range(len(body)) if re.match(sys.argv[3], body[i]))
if len(sys.argv) > 4:
    b = next(i for i in range(a + 1, len(body)) if re.match(sys.argv[4], body[i]))
seg = body[a:b]
print("instructions", len(seg), dict(Counter(cat(i) for i in seg)))
for c in ("dp", "valu32", "lds", "salu"):
    print(" ", c, Counter(i.split()[0] for i in seg if cat(i) == c).most_common(10))
