#!/usr/bin/env python3
"""How much of the RCCL point-to-point traffic of a rocprofv3 --kernel-trace run travels beside the assembly: for every
RCCL kernel (ncclDevKernel*) the time it overlaps k_ecsim_fill* launches on another stream.
usage: trace_overlap.py <rocprofv3 output dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
fill = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_ecsim_fill" in r["Kernel_Name"])
nccl = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Stream_Id", "?")) for r in rows if "nccl" in r["Kernel_Name"].lower()]
print("%d k_ecsim_fill launches, %d RCCL kernels" % (len(fill), len(nccl)))
big = sorted(nccl, key=lambda k: k[0] - k[1])[:12]
tot = ov = 0
for (a, b, name, st) in nccl:
    o = sum(max(0, min(b, fb) - max(a, fa)) for (fa, fb) in fill)
    tot += b - a
    ov += o
print("RCCL kernel time %.3f ms in all, %.3f ms of it (%.0f %%) beside k_ecsim_fill launches" % (tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1)))
print("the longest RCCL kernels (the matL ghost rows):")
for (a, b, name, st) in big:
    o = sum(max(0, min(b, fb) - max(a, fa)) for (fa, fb) in fill)
    print("  %-40s stream %s  %.3f ms, %.3f ms beside the assembly" % (name, st, (b - a) / 1e6, o / 1e6))
