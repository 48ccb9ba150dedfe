#!/usr/bin/env python3
"""One line of what a bench.py JSON line says about the step: tools/show_bench.py <file.json> [...]"""
import json
import sys

for f in sys.argv[1:]:
    l = json.loads(open(f).readline())
    p = l["phase_ms_per_step"]
    keys = ("fill_current", "solve_matA", "matA_apply", "precond", "second_push", "index", "scatter", "move_bin", "basic_push",
            "corr_first_push", "corr_second_push", "solve_matM")
    print("%s: %.2f ms/step, %.1f its, copy %.0f GB/s | " % (f, l["ms_per_step"], l.get("ksp_iterations_per_step") or 0, l.get("device_copy_GBps") or 0)
          + ", ".join("%s %.2f" % (k, p[k]) for k in keys if p.get(k)))
