#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of a rocprofv3 --pmc run (counter_collection.csv). usage: pmc_sq_summary.py <dir> [kernel substring]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_ecsim_fill"
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print("%-28s %18.0f  (%d launches)" % (k, agg[k], n[k]))
