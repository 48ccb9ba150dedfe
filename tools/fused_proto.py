#!/usr/bin/env python3
"""Times xpic_ecsim_second_push alone.  With tools/fused_proto.patch applied (git apply tools/fused_proto.patch) and a
build with EXTRA=-DSP_FUSED_PROTO=<slots per cell> the kernel is the one-pass re-binning prototype of DESIGN 5b (results
discarded: a measurement, not a code path); the product build gives the reference time.
usage: fused_proto.py [grid] [ppc] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = X.Context("ecsim", (n, n, n), (0.5,) * 3, 1.0)
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * n ** 3 * 1.02) + 1024)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
ctx.ecsim_second_push(s)
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(reps):
    ctx.ecsim_second_push(s)
ctx.synchronize()
nl, ms = ctx.profile_get("second_push")
print("second_push: %.3f ms per call (%d^3 x %d)" % (ms / nl, n, ppc))
