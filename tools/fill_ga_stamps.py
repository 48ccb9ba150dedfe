#!/usr/bin/env python3
"""Section shares of the GATHERING assembly inside whole ecsim steps (build with EXTRA="-DXPIC_EXPERIMENT -DFILL_STAMPS", run with
XPIC_ALLOW_EXPERIMENT=1): fill_ga_stamps.py [grid] [ppc] [steps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, 1.0)
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * n ** 3 * 1.02) + 1024)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
if len(sys.argv) > 4:
    ctx.set_fused_rebin(int(sys.argv[4]))
for _ in range(2):
    ctx.step()
st = np.zeros(16)
ctx.synchronize()
ctx.L.xpic_debug_fill_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 1)
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(steps):
    ctx.step()
ctx.synchronize()
ctx.L.xpic_debug_fill_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 1)
nl, ms = ctx.profile_get("fill_current")
names = ["loop top", "phase 1", "phase 2", "prefetch + RMW issue", "barrier 1", "seed + barrier 2", "merge + barrier 3", "flush", "barrier 4"]
tot = st[:9].sum()
print("assembly %.2f ms (%d launches of %.3f ms); section shares of wave 0, absolute = share x assembly time:" % (ms / steps, nl // steps, ms / nl))
for k, nm in enumerate(names):
    print("  %-22s %6.2f %%   %6.2f ms" % (nm, 100 * st[k] / tot, st[k] / tot * ms / steps))
for k, nm in ((9, "  of it: move + wrap (settle)"), (10, "  of it: the six sorted-copy stores"), (11, "  of it: gather issue (phase 1 + prefetch)"),
              (13, "  of it: the wait at the end of a pass"), (12, "  of it: the wait in front of the prefetch"),
              (14, "  of it: index prefetch (two cells ahead)"), (15, "  of it: RMW requests + offset table loads")):
    if st[k]:
        print("  %-42s %6.2f %%   %6.2f ms" % (nm, 100 * st[k] / tot, st[k] / tot * ms / steps))
