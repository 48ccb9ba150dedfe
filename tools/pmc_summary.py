#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs per kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section; re-checked here on k_copy8 / k_copy16 which move a known
1 GiB each way): the counters are in KiB; FETCH_SIZE reports exactly half of a coalesced streaming read for both
8- and 16-byte-per-lane loads, WRITE_SIZE is exact.  Output: corrected bytes per launch.
usage: pmc_summary.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run>
"""
import collections
import csv
import glob
import re
import sys


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpic_amd import csrc_hash
print("# csrc-hash %s   (bench.py quotes this file as roofline.traffic only while xpic_amd/csrc still hashes to this)" % csrc_hash())
print("%-28s %6s %12s %12s %12s %10s" % ("kernel", "calls", "read GB", "write GB", "total GB", "avg ms"))
for k in sorted(fetch, key=lambda k: -sum(v[1] for v in fetch[k])):
    fv, wv = fetch[k], write.get(k, [])
    rd = 2.0 * 1024 * sum(v[0] for v in fv) / len(fv) / 1e9
    wr = 1024 * sum(v[0] for v in wv) / max(len(wv), 1) / 1e9
    ms = sum(v[1] for v in fv) / len(fv)
    print("%-28s %6d %12.3f %12.3f %12.3f %10.3f" % (k, len(fv), rd, wr, rd + wr, ms))
