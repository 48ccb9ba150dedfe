#!/usr/bin/env python3
"""Residual history of the predict solve at the bench's parameters: |r_k| / |b| after k = 1 .. 6 outer iterations for the
preconditioner kinds given (the solve is cut off at k iterations; xpic_solve reports the norm it stopped at).
usage: resid_history.py [grid] [ppc] [kind[:degree] ...]      rhs = 2 E - dt currI + dt rot-(B - B0) of the second step"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kinds = sys.argv[3:] or ["3", "4"]
dt = 1.0
ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, dt)
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * n ** 3 * 1.02) + 1024)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
del B
ctx.step()  # fields with their thermal noise
ctx.ecsim_first_push(s)
ctx.update_cells(s)
ctx.ecsim_fill_current()
# rhs of advance_fields (ecsim/simulation.cpp:255-264) in W0: 2 E - dt currI + dt rot-(B - B0)
ctx.vec_axpby(X.W1, 0.0, 0.0, X.E)          # W1 = 0
ctx.vec_axpby(X.W1, 1.0, 1.0, X.B)          # W1 = B
ctx.vec_axpby(X.W1, -1.0, 1.0, X.B0)        # W1 = B - B0
ctx.vec_axpby(X.W0, 0.0, 0.0, X.E)
ctx.vec_axpby(X.W0, 2.0, 1.0, X.E)          # W0 = 2 E
ctx.vec_axpby(X.W0, -dt, 1.0, X.CURRI)      # - dt currI
ctx.rot_apply(-1, dt, X.W1, X.W0, add=True)
bn = ctx.vec_norm2(X.W0)
print("|b| = %.6e" % bn)
for spec in kinds:
    kind, _, deg = spec.partition(":")
    ctx.set_preconditioner(int(kind), int(deg or 0))
    out = []
    for k in range(1, 7):
        its, reason, rn = C.c_int(), C.c_int(), C.c_double()
        ctx.L.xpic_solve(ctx.h, X.OP_MATA_GMRES, X.W0, X.W2, C.c_double(1e-30), C.c_double(1e-300), k, C.byref(its), C.byref(reason), C.byref(rn))
        out.append(rn.value / bn)
    print("kind %-6s" % spec, " ".join("%.2e" % v for v in out), flush=True)
