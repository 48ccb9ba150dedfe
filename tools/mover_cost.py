#!/usr/bin/env python3
"""What particles that change cell cost k_scatter, by direction: thermal velocities along one axis only (or all, or
none), same speed distribution.  usage: mover_cost.py [grid] [ppc]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(3)
N = ppc * n ** 3
r = rng.random((N, 3)) * (0.5 * n)
for name, axes in (("none", ()), ("x only", (0,)), ("y only", (1,)), ("z only", (2,)), ("x, y, z", (0, 1, 2))):
    ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, 1.0)
    s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=N + 1024)
    v = np.zeros((N, 3))
    for a in axes:
        v[:, a] = rng.normal(0, 0.014 * np.sqrt(3.0 / max(len(axes), 1)), N)
    ctx.add_particles(s, np.hstack([r, v]))
    ctx.set_preconditioner(1)
    for _ in range(2):
        ctx.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(4):
        ctx.step()
    ctx.synchronize()
    out = {}
    for ph in ("scatter", "second_push"):
        nl, ms = ctx.profile_get(ph)
        out[ph] = round(ms / max(nl, 1), 3)
    print("%-8s movers: scatter %.3f ms, second_push %.3f ms per step" % (name, out["scatter"], out["second_push"]), flush=True)
    del ctx
