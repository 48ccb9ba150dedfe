#!/usr/bin/env python3
"""The assembly on ONE z-slab as a rank of an N-slab run sees it (ghost planes, non-periodic z inside the slab): a
single-slab context with geometry.self_ring.  usage: fill_slab.py [nx] [ny] [planes] [ppc] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

nx, ny, nz = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 256), (2, 256), (3, 32)))
ppc = int(sys.argv[4]) if len(sys.argv) > 4 else 64
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ctx = X.Context("ecsim", (nx, ny, nz), (0.5,) * 3, 1.0, self_ring=True)
ctx.comm_init_rccl(X.rccl_unique_id())  # its own lower and upper neighbour: the ghost-row exchange runs too (timed apart)
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * nx * ny * nz * 1.02) + 1024)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
ctx.ecsim_fill_current()
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(reps):
    ctx.ecsim_fill_current()
ctx.synchronize()
nl, ms = ctx.profile_get("fill_current")
print("slab %d x %d x %d, %d ppc: fill_current %.2f ms per assembly (%d colour launches, %.3f ms each)"
      % (nx, ny, nz, ppc, ms / reps, nl // reps, ms / nl))
