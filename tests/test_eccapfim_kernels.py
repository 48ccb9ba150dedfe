"""Inner kernels of the reference's `eccapfim` scheme: cell_traversal and ImplicitEsirkepov.

The reference's only fixture for this scheme is the `eccapfim_ex1` integration run, which needs the whole SNES /
Crank-Nicolson scheme: these kernels alone are **parity unpinned** against reference outputs.  What pins them here is
physics they must satisfy together with an already pinned kernel: the current deposited segment by segment obeys the
discrete continuity equation with the 2nd-order charge density of ChargeConservation (pinned by the golden
charge_conservation tables) to round-off, and the segments of cell_traversal tile the path exactly."""
import numpy as np
import pytest

N, D, DT = (9, 8, 7), (0.5, 0.4, 0.3), 0.7


def random_moves(rng, n, max_cells=1.4):
    L = np.array(N) * np.array(D)
    r0 = rng.random((n, 3)) * L * 0.6 + L * 0.2
    rn = r0 + (rng.random((n, 3)) * 2 - 1) * max_cells * np.array(D)
    return rn, r0


def segments(traverse, rn, r0):
    """[(segment end, segment start, fraction of the path)] per move, from cell_traversal's points."""
    out = []
    for q in range(rn.shape[0]):
        pts, cnt = traverse(rn[q], r0[q])
        assert cnt == len(pts)
        d = np.linalg.norm(rn[q] - r0[q])
        for s in range(1, cnt):
            ds = np.linalg.norm(pts[s] - pts[s - 1])
            out.append((q, pts[s], pts[s - 1], ds / d if d > 0 else 1.0))
    return out


def test_oracle_cell_traversal_tiles_the_path(oracle):
    o = oracle.OracleSim("basic", N, D, DT)
    rng = np.random.default_rng(3)
    rn, r0 = random_moves(rng, 200, 2.5)
    for q in range(200):
        pts, cnt = o.cell_traversal(rn[q], r0[q], 16)
        assert np.array_equal(pts[0], r0[q]) and np.array_equal(pts[-1], rn[q])
        # collinear, monotone, fractions sum to one
        t = (pts - r0[q]) @ (rn[q] - r0[q]) / np.dot(rn[q] - r0[q], rn[q] - r0[q])
        assert np.all(np.diff(t) >= -1e-15) and abs(t[-1] - 1) < 1e-14
        assert np.abs(pts - (r0[q] + np.outer(t, rn[q] - r0[q]))).max() < 1e-13
        # every segment stays inside one node-centred cell: its midpoint's cell is constant along it
        cells = np.round((0.5 * (pts[1:] + pts[:-1])) / np.array(D)).astype(int)
        assert len({tuple(c) for c in cells}) == len(cells) == cnt - 1
        # number of crossings = L1 distance of the end cells
        assert cnt - 2 == np.abs(np.round(rn[q] / D).astype(int) - np.round(r0[q] / D).astype(int)).sum()


def test_oracle_implicit_esirkepov_continuity(oracle):
    """div(-) J + (rho_new - rho_old) / dt = 0 to round-off for the segment-wise deposit of eccapfim/particles.cpp:
    decompose(q n/Np * ds/d, v = (rn - r0) / dt, segment)."""
    o = oracle.OracleSim("basic", N, D, DT)
    so = o.add_sort(3, 1.0, -1.0, 1.0)
    rng = np.random.default_rng(4)
    rn, r0 = random_moves(rng, 300)
    qn = -1.0 * 1.0 / 3
    o.add_particles(so, np.hstack([r0, np.zeros_like(r0)]))
    rho0 = o.charge_density(so)
    o.clear(so)
    o.add_particles(so, np.hstack([rn, np.zeros_like(rn)]))
    rho1 = o.charge_density(so)
    segs = segments(lambda a, b: o.cell_traversal(a, b, 16), rn, r0)
    v = (rn - r0) / DT
    o.set_field("J", np.zeros(o.fshape()))
    o.implicit_esirkepov_decompose([qn * f for (_, _, _, f) in segs], [v[q] for (q, _, _, _) in segs],
                                   [e for (_, e, _, _) in segs], [s for (_, _, s, _) in segs], "J")
    J = o.get_field("J")
    div = np.zeros(rho0.shape)
    oracle.lib().orc_div_neg(o.h, oracle._dp(np.ascontiguousarray(J)), oracle._dp(div))
    res = (rho1 - rho0) / DT + div
    assert np.abs(div).max() > 0
    assert np.abs(res).max() <= 1e-12 * np.abs(div).max()


def test_oracle_implicit_esirkepov_interpolates_constants(oracle):
    """The 54 weights of a segment sum to one per component, so a uniform E comes back exactly; B likewise."""
    o = oracle.OracleSim("basic", N, D, DT)
    E = np.zeros(o.fshape()) + np.array([0.3, -1.1, 0.7])
    B = np.zeros(o.fshape()) + np.array([-0.2, 0.5, 0.9])
    o.set_field("E", E)
    o.set_field("B", B)
    rng = np.random.default_rng(5)
    rn, r0 = random_moves(rng, 100, 0.45)  # segments inside one cell
    Ep, Bp = o.implicit_esirkepov_interpolate(rn, r0)
    assert np.abs(Ep - np.array([0.3, -1.1, 0.7])).max() < 1e-13
    assert np.abs(Bp - np.array([-0.2, 0.5, 0.9])).max() < 1e-13


@pytest.mark.gpu
def test_device_kernels_match_oracle(oracle):
    import xpic_amd as X

    o = oracle.OracleSim("basic", N, D, DT)
    g = X.Context("basic", N, D, DT)
    rng = np.random.default_rng(6)
    E, B = rng.normal(0, 1, o.fshape()), rng.normal(0, 1, o.fshape())
    for name, fid, F in (("E", X.E, E), ("B", X.B, B)):
        o.set_field(name, F)
        g.set_field(fid, F)
    rn, r0 = random_moves(rng, 500, 2.2)
    # cell_traversal: the same points up to the device's fused multiply-add in start + dir * t (a few ulp)
    pts, counts = g.cell_traversal(rn, r0, 16)
    for q in range(rn.shape[0]):
        po, cnt = o.cell_traversal(rn[q], r0[q], 16)
        assert counts[q] == cnt
        assert np.abs(pts[q, :cnt] - po).max() <= 4 * np.finfo(float).eps * np.abs(po).max()
    # interpolate / decompose on the segments
    segs = segments(lambda a, b: o.cell_traversal(a, b, 16), rn[:200], r0[:200])
    se = np.array([e for (_, e, _, _) in segs])
    ss = np.array([s for (_, _, s, _) in segs])
    Eo, Bo = o.implicit_esirkepov_interpolate(se, ss)
    Eg, Bg = g.implicit_esirkepov_interpolate(se, ss)
    assert np.abs(Eo - Eg).max() <= 1e-13 * np.abs(Eo).max()
    assert np.abs(Bo - Bg).max() <= 1e-13 * np.abs(Bo).max()
    alpha = rng.normal(0, 1, len(segs))
    v = rng.normal(0, 1, (len(segs), 3))
    o.set_field("J", np.zeros(o.fshape()))
    g.vec_set(X.J, 0.0)
    o.implicit_esirkepov_decompose(alpha, v, se, ss, "J")
    g.implicit_esirkepov_decompose(alpha, v, se, ss, X.J)
    a, b = o.get_field("J"), g.get_field(X.J)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
