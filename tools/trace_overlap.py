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

# round 5: whatever else runs beside the assembly -- with the ghost rows on the copy path no RCCL kernel should, and the
# copies themselves show up as memory-copy records (between two GPUs: SDMA engines; on one GPU, as here on a self-ring, the
# runtime may run them as blit kernels, listed below if it does)
other = {}
for r in rows:
    if "k_ecsim_fill" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower():
        continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    o = sum(max(0, min(b, fb) - max(a, fa)) for (fa, fb) in fill)
    if o > 0:
        k = r["Kernel_Name"][:60]
        other[k] = other.get(k, 0) + o
print("other kernels that overlap k_ecsim_fill launches (ms of overlap):", {k: round(v / 1e6, 3) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:8]})
mc = glob.glob(sys.argv[1] + "/*/*memory_copy_trace.csv")
if mc:
    cps = list(csv.DictReader(open(mc[0])))
    tot = ov = 0
    n = 0
    for r in cps:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b - a < 100000:  # (the ghost-row planes are 65 MB each: tens of microseconds at least; skip the small bookkeeping copies)
            continue
        n += 1
        tot += b - a
        ov += sum(max(0, min(b, fb) - max(a, fa)) for (fa, fb) in fill)
    print("%d large memory copies, %.3f ms in all, %.3f ms of it beside k_ecsim_fill launches" % (n, tot / 1e6, ov / 1e6))
