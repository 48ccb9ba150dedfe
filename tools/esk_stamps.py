#!/usr/bin/env python3
"""Section shares of the Esirkepov push kernels (build with EXTRA="-DXPIC_EXPERIMENT -DESK_STAMPS", run with XPIC_ALLOW_EXPERIMENT=1): esk_stamps.py [grid] [ppc]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
names = ["flush + shift (loop top)", "tile / nb -> LDS", "compose", "barrier A", "request", "phase 1", "stage + slow",
         "barrier B (+tile)", "phase 2", "-"]
for scheme, dt in (("basic", 0.1), ("ecsimcorr", 1.0)):
    ctx = X.Context(scheme, (n, n, n), (0.5,) * 3, dt)
    s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * n ** 3 * 1.02) + 1024)
    ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
    B = np.zeros(ctx.fshape())
    B[..., 2] = 0.2
    ctx.set_field(X.B, B)
    ctx.set_field(X.B0, B)
    ctx.step()
    st = np.zeros(16)
    ctx.synchronize()
    ctx.L.xpic_debug_esk_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 1)
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.step()
    ctx.synchronize()
    ctx.L.xpic_debug_esk_stamps(st.ctypes.data_as(C.POINTER(C.c_double)), 1)
    tot = st[:10].sum()
    print(scheme, "(all Esirkepov kernels of one step; wave 0 of every workgroup)")
    for k, nm in enumerate(names):
        print("  %-26s %6.2f %%" % (nm, 100 * st[k] / tot))
    for ph in ("basic_push", "corr_first_push", "corr_second_push"):
        nl, ms = ctx.profile_get(ph)
        if nl:
            print("  %s: %.2f ms" % (ph, ms / nl))
    del ctx
