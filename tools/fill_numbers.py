#!/usr/bin/env python3
"""Fills the @@NAME@@ placeholders of DESIGN.md / README.md from the committed bench lines: tools/fill_numbers.py [round]"""
import csv
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rd = sys.argv[1] if len(sys.argv) > 1 else "r04"
l = json.load(open("%s/profiles/%s_bench_256.json" % (R, rd)))
ph = l["phase_ms_per_step"]
side = {k: json.load(open("%s/profiles/%s_bench_%s.json" % (R, rd, f)))["ms_per_step"]
        for k, f in (("BASIC", "basic_128"), ("CORR", "ecsimcorr_128"), ("CFG4", "cfg4_512x512x64"))}
prof = None
for r in csv.DictReader(open("%s/profiles/%s_rocprofv3_kernel_stats_256.csv" % (R, rd))):
    if "k_ecsim_fill<" in r["Name"]:
        prof = float(r["AverageNs"]) / 1e6
v = {"MS": "%.1f" % l["ms_per_step"], "PPS": "%.2f" % (l["value"] / 1e9), "ITS": "%.1f" % l["ksp_iters_per_s"],
     "FILL": "%.1f" % ph["fill_current"], "FILLAVG": "%.2f" % l["roofline"]["avg_ms"], "FRAC": "%.3f" % l["roofline"]["frac"],
     "FILLPROF": "%.2f" % prof, "SOLVE": "%.1f" % ph["solve_matA"], "MATA": "%.1f" % ph["matA_apply"], "PREC": "%.1f" % ph["precond"],
     "PUSH": "%.1f" % ph["second_push"]}
v.update({k: "%.1f" % x for k, x in side.items()})
for f in ("DESIGN.md", "README.md"):
    s = open(os.path.join(R, f)).read()
    for k, x in v.items():
        s = s.replace("@@%s@@" % k, x)
    assert "@@" not in s, f
    open(os.path.join(R, f), "w").write(s)
print(v)
