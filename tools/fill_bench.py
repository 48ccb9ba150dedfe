#!/usr/bin/env python3
"""Times xpic_ecsim_fill_current alone (no solve: usable with the FILL_EXP builds whose results are garbage).
usage: fill_bench.py [grid] [ppc] [reps] [kernels]     kernels: e.g. "1 0 1 0" = A/B/A/B of the warp-specialised (1) and classic (0) body"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, 1.0)
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * n ** 3 * 1.02) + 1024)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
kinds = [int(k) for k in sys.argv[4].split()] if len(sys.argv) > 4 else [1]
ctx.ecsim_fill_current()
ctx.profile_enable(True)
for kind in kinds[:-1]:
    ctx.set_fill_kernel(kind)
    ctx.ecsim_fill_current()
    ctx.profile_reset()
    for _ in range(reps):
        ctx.ecsim_fill_current()
    nl, ms = ctx.profile_get("fill_current")
    print("kernel %d %s: %.2f ms per assembly (%d colour launches, %.3f ms each)" % (kind, ctx.fill_variant(), ms / reps, nl // reps, ms / nl), flush=True)
ctx.set_fill_kernel(kinds[-1])
ctx.ecsim_fill_current()
ctx.profile_reset()
for _ in range(reps):
    ctx.ecsim_fill_current()
if hasattr(ctx.L, "xpic_debug_fill_stamps"):
    import ctypes as C
    st = np.zeros(16)
    ctx.synchronize()
    ctx.L.xpic_debug_fill_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 0)
    names = ["loop top", "phase 1", "phase 2", "prefetch + RMW issue", "barrier 1", "seed + barrier 2",
             "merge + barrier 3", "flush", "barrier 4"]
    tot = st[:9].sum()
    if tot > 0:
        print("section shares of wave 0 (s_memtime ticks, all workgroups, warm-up included):")
        for k, nm in enumerate(names):
            print("  %-22s %6.2f %%" % (nm, 100 * st[k] / tot))
if hasattr(ctx.L, "xpic_debug_fill_ws_stamps") and kinds[-1] == 1:
    import ctypes as C
    st = np.zeros(32)
    ctx.synchronize()
    ctx.L.xpic_debug_fill_ws_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 0)
    cons = ["wait FULL", "header", "phase 2", "post FREE", "-", "wait SEEDED", "merge + bump", "-"]
    for base, role, names in ((0, "consumer oz = 0 (wave 0)", cons), (8, "consumer oz = 1 (wave 4)", cons),
                              (16, "producer (wave 8)", ["wait FREE", "-", "cell prologue / prefetch", "phase 1", "header + next loads + post FULL", "-", "-", "-"]),
                              (24, "flusher (wave 12)", ["wait MERGED", "flush + seed", "request old", "tail", "-", "-", "-", "-"])):
        tot = st[base:base + 8].sum()
        print("section shares of the %s, %.3g ticks (all workgroups, warm-up included):" % (role, tot))
        for k, nm in enumerate(names):
            if st[base + k]:
                print("  %-34s %6.2f %%" % (nm, 100 * st[base + k] / tot))
nl, ms = ctx.profile_get("fill_current")
print("kernel %d %s: " % (kinds[-1], ctx.fill_variant()), end="")
print("fill_current: %.2f ms per assembly (%d colour launches, %.3f ms each)" % (ms / reps, nl // reps, ms / nl))
