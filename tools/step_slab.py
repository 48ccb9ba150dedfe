#!/usr/bin/env python3
"""One rank's share of an N-slab run on ONE GPU: a single slab of nz / N planes that keeps its ghost planes and is its
own neighbour (geometry.self_ring; the RCCL ring sends to itself).  What it shows: how the per-rank work scales down
(launch granularity, colour schedule, fixed costs), NOT the links.  usage: step_slab.py [scheme] [nx] [ny] [planes] [ppc]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xpic_amd as X

scheme = sys.argv[1] if len(sys.argv) > 1 else "ecsim"
nx, ny, nz = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((2, 256), (3, 256), (4, 32)))
ppc = int(sys.argv[5]) if len(sys.argv) > 5 else 64
steps = 5
ctx = X.Context(scheme, (nx, ny, nz), (0.5,) * 3, 1.0, self_ring=True)
ctx.comm_init_rccl(X.rccl_unique_id())
if os.environ.get("XPIC_SLAB_PEER") == "1":  # the copy-engine path needs the neighbours' buffers mapped: its own, on a self-ring
    blob = ctx.comm_peer_export()
    ctx.comm_peer_import(blob, blob)
if os.environ.get("XPIC_SLAB_OVERLAP") is not None:  # 0: every exchange blocks the compute stream (A/B against the default); 4: ghost rows by peer copy
    ctx.set_overlap(int(os.environ["XPIC_SLAB_OVERLAP"]))
s = ctx.add_sort(ppc, 1.0, -1.0, 1.0, capacity=int(ppc * nx * ny * nz * 1.05) + 4096)
ctx.fill_synthetic(s, ppc, 0.014, seed=1234)
B = np.zeros(ctx.fshape())
B[..., 2] = 0.2
ctx.set_field(X.B, B)
ctx.set_field(X.B0, B)
for _ in range(2):
    ctx.step()
ctx.synchronize()
ctx.profile_enable(True)
ctx.profile_reset()
ctx.comm_stats(reset=True)
import time
t0 = time.perf_counter()
its = 0
for _ in range(steps):
    its += ctx.step()
ctx.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
names = ("fill_current", "solve_matA", "solve_matM", "matA_apply", "precond", "scatter", "second_push", "move_bin", "halo",
         "migrate", "index", "matL_ghost_rows", "matL_zero", "precond_setup", "rot_apply", "mdot", "maxpy", "corr_first_push", "corr_second_push", "matL_apply", "scan", "allreduce", "peer_copies")
prof = {k: ctx.profile_get(k) for k in names}
print("%s slab %d x %d x %d, %d ppc (self-ring): %.2f ms/step, %.1f iterations/step" % (scheme, nx, ny, nz, ppc, ms, its / steps))
print("  " + ", ".join("%s %.2f" % (k, v[1] / steps) for k, v in prof.items() if v[1] / steps > 0.05),
      "| all-reduces/step %.1f" % (prof["allreduce"][0] / steps))
cs = ctx.comm_stats()
print("  per step on the links: %.1f messages, %.1f MB sent, %.1f all-reduces of %.0f B in all" % (cs[0] / steps, cs[1] / steps / 1e6, cs[2] / steps, cs[3] / steps))
