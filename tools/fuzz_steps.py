#!/usr/bin/env python3
"""Random small configurations (grid extents, spacings, particles per cell from 0 to a few hundred, clustered cells),
two steps of every scheme on the GPU against the CPU oracle.  usage: fuzz_steps.py [cases] [seed] [largest nx + 1] [slab]
"slab": the device side is a single z-slab that keeps its ghost planes and is its own neighbour over RCCL
(geometry.self_ring) -- the slab code paths (ghost exchanges, migration, colour schedule, cleared boundary planes of the
assembly) on the same random configurations, empty stretches and heavy cells included."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import xpic_amd as X



def run(cases=12, seed=7, verbose=True, nxmax=41, slab=False):
    rng = np.random.default_rng(seed)
    oracle_lib.lib()
    worst = {}
    for case in range(cases):
        _case(case, rng, worst, verbose, nxmax, slab)
    return worst


def _case(case, rng, worst, verbose, nxmax, slab=False):
    scheme = ("basic", "ecsim", "ecsimcorr")[case % 3]
    n = tuple(int(v) for v in rng.integers(6, [nxmax, 13, 12]))
    d = tuple(float(v) for v in rng.choice([0.25, 0.4, 0.5, 0.7], 3))
    dt = float(rng.choice([0.05, 0.1]) if scheme == "basic" else rng.choice([0.2, 0.5]))
    o = oracle_lib.OracleSim(scheme, n, d, dt)
    g = X.Context(scheme, n, d, dt, self_ring=slab)
    if slab:
        g.comm_init_rccl(X.rccl_unique_id())
    if scheme != "basic":
        g.set_preconditioner(int(rng.integers(0, 5)))  # none, polynomial in matM (fp32 / fp64 vectors), in matM + <matL> (+ density-scaled rows)
    # Poisson background + a few heavy cells + empty stretches
    ppc = float(rng.choice([0.3, 3.0, 20.0, 70.0]))
    cnt = rng.poisson(ppc, n[::-1])
    cnt[:, :, : n[0] // 3] = 0 if rng.random() < 0.3 else cnt[:, :, : n[0] // 3]
    for _ in range(3):
        cz, cy, cx = (int(rng.integers(0, m)) for m in n[::-1])
        cnt[cz, cy, cx] = int(rng.choice([1, 64, 65, 240, 241, 300, 700]))
    pts = []
    for (cz, cy, cx), c in np.ndenumerate(cnt):
        if c:
            r = (np.array([cx, cy, cz]) + rng.random((c, 3))) * np.array(d)
            pts.append(np.hstack([r, np.clip(rng.normal(0, 0.25, (c, 3)), -0.6 * min(d) / dt, 0.6 * min(d) / dt)]))
    pts = np.vstack(pts) if pts else np.zeros((0, 6))
    o.add_sort(10, 1.0, -1.0, 1.0)
    g.add_sort(10, 1.0, -1.0, 1.0, capacity=len(pts) + 1000)
    if len(pts):
        assert o.add_particles(0, pts) == g.add_particles(0, pts) == len(pts)
    B = np.zeros(o.fshape()) + rng.normal(0, 0.2, 3)
    for name, fid in (("B", X.B), ("B0", X.B0)):
        o.set_field(name, B)
        g.set_field(fid, B)
    for sim in (o, g):
        sim.set_tolerances(1e-11, 1e-50, 400)
    for t in range(2):
        if o.step() < 0:
            # a particle moved further than the reference's Shape::shape[] holds (the fields of the heavy cells can
            # accelerate one that far): the oracle refuses the step, and so must the device
            try:
                g.step()
            except X.XpicError as e:
                assert "moved more than one cell" in str(e), (case, scheme, n, str(e))
                if verbose:
                    print("case %2d %-9s n=%-14s both sides refuse the step (move of more than a cell)" % (case, scheme, n), flush=True)
                return
            raise AssertionError((case, scheme, n, "the oracle refused the step, the device did not"))
        g.step()
    err = 0.0
    for name, fid in (("E", X.E), ("B", X.B)):
        a, b = o.get_field(name), g.get_field(fid)
        err = max(err, np.abs(a - b).max() / max(np.abs(a).max(), 1e-300))
    po, co = o.particles(0)
    pg, cg = g.particles(0)
    assert len(po) == len(pg), (case, scheme, n)
    if len(po):
        io, ig = np.lexsort(po.T[::-1]), np.lexsort(pg.T[::-1])
        perr = np.abs(po[io] - pg[ig]).max()
        assert np.array_equal(np.sort(co), np.sort(cg)), (case, scheme, n, "cells differ")
    else:
        perr = 0.0
    if verbose:
        print("case %2d %-9s n=%-14s d=%-18s ppc~%-5g N=%-6d field err %.1e particle err %.1e" % (case, scheme, n, d, ppc, len(pts), err, perr), flush=True)
    assert err < 1e-6 and perr < 1e-8, (case, scheme, n)
    worst[scheme] = max(worst.get(scheme, 0.0), err)
    del g

if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 7,
            nxmax=int(sys.argv[3]) if len(sys.argv) > 3 else 41, slab=len(sys.argv) > 4 and sys.argv[4] == "slab")
    print("worst relative field error per scheme:", w)
