"""Edge cases through the C ABI, against the CPU oracle: empty and ragged species, cells holding far more particles
than one staging pass, grid extents that are not multiples of any tile size, a single particle."""
import numpy as np
import pytest

from test_gpu_ecsim import canon

pytestmark = pytest.mark.gpu


def pair(oracle, scheme, n, d, dt):
    import xpic_amd as X

    o = oracle.OracleSim(scheme, n, d, dt)
    g = X.Context(scheme, n, d, dt)
    if scheme != "basic":
        g.set_preconditioner(0)
    return o, g


def same_fields(o, g, names, tol=1e-9):
    import xpic_amd as X

    ids = {"E": X.E, "B": X.B, "J": X.J, "currI": X.CURRI}
    for name in names:
        a, b = o.get_field(name), g.get_field(ids[name])
        scale = max(np.abs(a).max(), 1e-300)
        assert np.abs(a - b).max() <= tol * scale, name


@pytest.mark.parametrize("scheme", ["ecsim", "ecsimcorr", "basic"])
def test_empty_species(oracle, scheme):
    """No particles at all: the step is a vacuum Maxwell update, nothing may fault or hang."""
    import xpic_amd as X

    n, d = (8, 7, 6), (0.5, 0.5, 0.5)
    o, g = pair(oracle, scheme, n, d, 0.3)
    o.add_sort(4, 1.0, -1.0, 1.0)
    g.add_sort(4, 1.0, -1.0, 1.0, capacity=1000)
    rng = np.random.default_rng(1)
    E, B = rng.normal(0, 0.1, o.fshape()), rng.normal(0, 0.1, o.fshape())
    for name, fid, F in (("E", X.E, E), ("B", X.B, B)):
        o.set_field(name, F)
        g.set_field(fid, F)
    for _ in range(2):
        o.step()
        g.step()
    same_fields(o, g, ["E", "B"], 1e-7)
    assert g.count(0) == 0
    assert g.update_cells(0) == 0


def test_ragged_cells_and_long_cells(oracle):
    """Vacuum almost everywhere, one cell with 300 particles (10 staging passes), one with 33, one with 1."""
    import xpic_amd as X

    n, d = (13, 9, 7), (0.5, 0.4, 0.3)
    o, g = pair(oracle, "ecsim", n, d, 0.5)
    o.add_sort(10, 1.0, -1.0, 1.0)
    g.add_sort(10, 1.0, -1.0, 1.0, capacity=5000)
    rng = np.random.default_rng(2)
    pts = []
    for cell, cnt in (((12, 8, 6), 300), ((0, 0, 0), 33), ((5, 3, 2), 1), ((6, 3, 2), 64), ((12, 0, 6), 65)):
        r = (np.array(cell) + rng.random((cnt, 3))) * np.array(d)
        pts.append(np.hstack([r, rng.normal(0, 0.2, (cnt, 3))]))
    pts = np.vstack(pts)
    assert o.add_particles(0, pts) == g.add_particles(0, pts) == len(pts)
    B = np.zeros(o.fshape()) + np.array([0.1, -0.2, 0.4])
    for name, fid in (("B", X.B), ("B0", X.B0)):
        o.set_field(name, B)
        g.set_field(fid, B)
    oracle.lib().orc_ecsim_fill_current(o.h)
    g.ecsim_fill_current()
    Lo, Lg = o.matL(), g.matL()
    assert np.abs(Lo - Lg).max() <= 1e-12 * np.abs(Lo).max()
    assert np.count_nonzero(Lg) == np.count_nonzero(Lo)  # untouched rows are exactly zero
    same_fields(o, g, ["currI"], 1e-12)
    for s in (o, g):
        s.set_tolerances(1e-11, 1e-50, 300)
    for _ in range(3):
        assert o.step() >= 0
        g.step()
    same_fields(o, g, ["E", "B"], 1e-6)
    po, co = canon(*o.particles(0))
    pg, cg = canon(*g.particles(0))
    assert np.array_equal(co, cg) and np.abs(po - pg).max() < 1e-9


@pytest.mark.parametrize("d", [(0.5, 0.4, 0.3), (0.5, 0.25, 0.5)], ids=["general", "pow2"])
def test_assembly_kernels_on_pass_boundaries(oracle, d):
    """Both assembly kernels (xpic_set_fill_kernel: 1 = warp-specialised producer / consumer waves with 44-slot stage
    buffers, 0 = classic 64-slot passes) on one pencil whose cells hold 0, 1, 3, 4, 5, 43, 44, 45, 47, 48, 63, 64, 65,
    87, 88, 89, 131, 133 and 700 particles -- every way a cell can end on, just before or just after a pass or a K = 4
    step -- with vacuum in between, two species, B != 0: currI and matL equal the oracle's to 1e-12 for both."""
    import xpic_amd as X

    n = (24, 7, 6)
    o, g = pair(oracle, "ecsim", n, d, 0.5)
    rng = np.random.default_rng(12)
    counts = [0, 1, 3, 4, 5, 43, 44, 45, 47, 48, 63, 64, 65, 87, 88, 89, 131, 133, 700, 0, 2, 46, 90, 7]
    for sp, (q, m) in enumerate(((-1.0, 1.0), (1.0, 30.0))):
        o.add_sort(10, 1.0, q, m)
        g.add_sort(10, 1.0, q, m, capacity=20000)
        pts = []
        for cx, cnt in enumerate(counts if sp == 0 else counts[::-1]):
            for (cy, cz) in ((3, 2), (4, 2), (0, 5)):  # two y-neighbouring pencils and one across the periodic z edge
                c = cnt if (cy, cz) == (3, 2) else cnt // 3
                if c:
                    r = (np.array([cx, cy, cz]) + rng.random((c, 3))) * np.array(d)
                    pts.append(np.hstack([r, rng.normal(0, 0.2, (c, 3))]))
        pts = np.vstack(pts)
        assert o.add_particles(sp, pts) == g.add_particles(sp, pts) == len(pts)
    B = rng.normal(0, 0.3, o.fshape()) + np.array([0.1, -0.2, 0.4])
    for name, fid in (("B", X.B), ("B0", X.B0)):
        o.set_field(name, B)
        g.set_field(fid, B)
    oracle.lib().orc_ecsim_fill_current(o.h)
    Lo = o.matL()
    g.set_fill_kernel(1)
    assert g.fill_variant() == (d[1] == 0.25, True, True)
    for kind in (1, 0, 1):
        g.set_fill_kernel(kind)
        assert g.fill_variant()[2] == bool(kind)
        g.ecsim_fill_current()
        Lg = g.matL()
        assert np.abs(Lo - Lg).max() <= 1e-12 * np.abs(Lo).max(), kind
        assert np.count_nonzero(Lg) == np.count_nonzero(Lo), kind
        same_fields(o, g, ["currI"], 1e-12)
        for sp in range(2):
            a, b = o.sort_current(sp, "currI"), g.sort_current(sp, X.CURRI)
            assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max(), (kind, sp)


def test_single_particle_basic(oracle):
    import xpic_amd as X

    n, d = (7, 6, 9), (0.5, 0.5, 0.5)
    o, g = pair(oracle, "basic", n, d, 0.1)
    o.add_sort(1, 1.0, -1.0, 1.0)
    g.add_sort(1, 1.0, -1.0, 1.0, capacity=16)
    p = np.array([[3.49999, 0.0, 4.25, 0.3, -0.8, 0.9]])  # sits on a cell face in y, about to cross in x
    assert o.add_particles(0, p) == g.add_particles(0, p) == 1
    for _ in range(12):
        assert o.step() == 0
        g.step()
    same_fields(o, g, ["E", "B", "J"], 1e-10)
    assert np.abs(o.particles(0)[0] - g.particles(0)[0]).max() < 1e-12
    assert o.particles(0)[1][0] == g.particles(0)[1][0]


def test_capacity_and_argument_errors():
    import xpic_amd as X

    g = X.Context("ecsim", (8, 8, 8), (0.5, 0.5, 0.5), 1.0)
    s = g.add_sort(1, 1.0, -1.0, 1.0, capacity=10)
    with pytest.raises(X.XpicError, match="capacity"):
        g.add_particles(s, np.zeros((11, 6)) + 1.0)
    with pytest.raises(X.XpicError):
        g.set_field(99, np.zeros(g.fshape()))
    with pytest.raises(X.XpicError, match="in place"):
        g.rot_apply(+1, 1.0, X.E, X.E)
    with pytest.raises(X.XpicError):
        X.Context("ecsim", (1, 8, 8), (0.5, 0.5, 0.5), 1.0)  # a CIC footprint cannot fold onto a single cell
    with pytest.raises(X.XpicError):
        X.Context("ecsimcorr", (3, 8, 8), (0.5, 0.5, 0.5), 1.0)  # extent below the 2nd-order shapes' width
    with pytest.raises(X.XpicError, match="basic scheme"):
        X.Context("basic", (8, 8, 8), (0.5, 0.5, 0.5), 1.0).ecsim_fill_current()


def test_long_pencils_and_odd_colour_periods(oracle):
    """A pencil longer than the LDS cell_start row (1024 cells), extents whose colour periods are 5 and 4, nx not a
    multiple of the 4-wide matL x-block, two species: assembly, solve and push against the oracle for two steps."""
    import xpic_amd as X

    n, d = (1030, 5, 4), (0.5, 0.5, 0.5)
    o, g = pair(oracle, "ecsim", n, d, 0.4)
    rng = np.random.default_rng(11)
    N = n[0] * n[1] * n[2]
    for (q, m) in ((-1.0, 1.0), (1.0, 25.0)):
        so = o.add_sort(2, 1.0, q, m)
        sg = g.add_sort(2, 1.0, q, m, capacity=4 * N)
        pts = np.hstack([rng.random((2 * N, 3)) * np.array(n) * np.array(d), rng.normal(0, 0.05, (2 * N, 3))])
        assert o.add_particles(so, pts) == g.add_particles(sg, pts) == 2 * N
    B = np.zeros(o.fshape()) + np.array([0.05, 0.1, -0.2])
    for name, fid in (("B", X.B), ("B0", X.B0)):
        o.set_field(name, B)
        g.set_field(fid, B)
    oracle.lib().orc_ecsim_fill_current(o.h)
    g.ecsim_fill_current()
    Lo, Lg = o.matL(), g.matL()
    assert np.abs(Lo - Lg).max() <= 1e-12 * np.abs(Lo).max()
    same_fields(o, g, ["currI"], 1e-12)
    for s in (o, g):
        s.set_tolerances(1e-11, 1e-50, 300)
    for _ in range(2):
        assert o.step() >= 0
        g.step()
    same_fields(o, g, ["E", "B"], 1e-6)
    for k in range(2):
        po, co = canon(*o.particles(k))
        pg, cg = canon(*g.particles(k))
        assert np.array_equal(co, cg) and np.abs(po - pg).max() < 1e-9


@pytest.mark.parametrize("table", ["precomposed", "overflows"])
@pytest.mark.parametrize("scheme", ["basic", "ecsimcorr"])
def test_esirkepov_rounds_ragged_pencils(oracle, scheme, table):
    """The Esirkepov kernels pack the particles of up to 8 consecutive cells into one round of 240 (160) stage columns:
    a cell of 700 particles (several rounds of its own, the rest sharing a round with its neighbours), one of exactly
    240, runs of empty cells longer than a round's 8 cells, cells of 1..5 particles (all padding), a pencil that ends in
    a full cell, and particles fast enough to leave the 4-node box (the cooperative slow path).  The rounds come out of the
    table k_esk_rounds composes beforehand; `overflows` adds a cell of 8 000 particles -- more rounds than the table holds for
    its pencil (nx + 11 = 32), so the pushes return at once and are launched again in the form that composes for itself."""
    import xpic_amd as X

    n, d = (21, 7, 6), (0.5, 0.4, 0.5)
    dt = 0.05 if scheme == "basic" else 0.2
    o, g = pair(oracle, scheme, n, d, dt)
    o.add_sort(10, 1.0, -1.0, 1.0)
    g.add_sort(10, 1.0, -1.0, 1.0, capacity=30000)
    rng = np.random.default_rng(5)
    pts = []
    cells = ([((7, 4, 3), 8000)] if table == "overflows" else []) + [((3, 2, 1), 700), ((4, 2, 1), 37), ((5, 2, 1), 240), ((20, 2, 1), 130), ((0, 2, 1), 3),
             ((19, 6, 5), 64), ((20, 6, 5), 65), ((0, 0, 0), 1), ((11, 3, 3), 5), ((12, 3, 3), 2), ((13, 3, 3), 241)]
    for cell, cnt in cells:
        r = (np.array(cell) + rng.random((cnt, 3))) * np.array(d)
        v = rng.normal(0, 0.3, (cnt, 3))
        v[: max(1, min(cnt, 1000) // 16)] *= 6.0  # a few that move most of a cell in a step
        v = np.clip(v, -0.7 * min(d) / dt, 0.7 * min(d) / dt)  # beyond ~0.8 cell the box of a move exceeds Shape::shape[]
        pts.append(np.hstack([r, v]))
    pts = np.vstack(pts)
    assert o.add_particles(0, pts) == g.add_particles(0, pts) == len(pts)
    B = np.zeros(o.fshape()) + np.array([0.1, -0.2, 0.3])
    for name, fid in (("B", X.B), ("B0", X.B0)):
        o.set_field(name, B)
        g.set_field(fid, B)
    for sim in (o, g):
        sim.set_tolerances(1e-11, 1e-50, 300)
    g.profile_enable(True)
    for t in range(3 if table == "precomposed" else 2):  # (the blob's own field throws particles further than a cell in its third step)
        assert o.step() >= 0
        g.profile_reset()
        g.step()
        # one launch of every push per step from the table; two (the first returned at once) when a pencil overflows it
        for ph in (("basic_push",) if scheme == "basic" else ("corr_first_push", "corr_second_push")):
            # (later steps: the blob has spread along its pencil and may fit)
            assert g.profile_get(ph)[0] in ((1,) if table == "precomposed" else ((2,) if t == 0 else (1, 2))), (t, ph)
        # ... and the pushes that composed their rounds themselves are counted for the caller to see
        nself = g.profile_get("esk_self_composed")[0]
        assert nself == 0 if table == "precomposed" else nself >= (1 if scheme == "basic" else 2) * (1 if t == 0 else 0), (t, nself)
        same_fields(o, g, ["E", "B"], 1e-7)
    po, co = canon(*o.particles(0))
    pg, cg = canon(*g.particles(0))
    assert np.array_equal(co, cg) and np.abs(po - pg).max() < 1e-9


@pytest.mark.parametrize("n", [(2, 2, 32), (3, 2, 5), (2, 3, 4), (5, 3, 2)])
def test_tiny_extents_like_the_reference_default_config(oracle, n):
    """The reference's shipped config.json is a 2 x 2 x 32 quasi-1D box (config.json:5-7): extents of 2 and 3 cells, where
    a CIC footprint wraps onto itself.  ecsim assembly (currI, matL through its SpMV), solve and two full steps against
    the oracle on the same particles."""
    import xpic_amd as X
    from test_gpu_ecsim import make_pair, canon

    o, g = make_pair(oracle, "ecsim", n, (0.5, 0.4, 0.25), 0.7, [(8, 1.0, -1.0, 1.0)], ppc=9, B0=(0.1, 0.0, 0.3), vth=0.05)
    oracle.lib().orc_ecsim_fill_current(o.h)
    g.ecsim_fill_current()
    a, b = o.get_field("currI"), g.get_field(X.CURRI)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    x = np.random.default_rng(5).normal(0, 1, o.fshape())
    g.set_field(X.W0, x)
    g.matL_apply(X.W0, X.W1)
    ya, yb = o.matL_apply(x), g.get_field(X.W1)
    assert np.abs(ya - yb).max() <= 1e-12 * np.abs(ya).max()
    for s_ in (o, g):
        s_.set_tolerances(1e-12, 1e-50, 400)
    for t in range(2):
        io, ig = o.step(), g.step()
        assert io > 0 and abs(io - ig) <= 2, (t, io, ig)
        for name, fid in (("E", X.E), ("B", X.B)):
            fa, fb = o.get_field(name), g.get_field(fid)
            assert np.abs(fa - fb).max() <= 1e-8 * np.abs(fa).max(), (t, name)
    po, co = canon(*o.particles(0))
    pg, cg = canon(*g.particles(0))
    assert np.array_equal(co, cg)
    assert np.abs(po - pg).max() <= 1e-11  # two field solves in: velocities (and the positions they move) agree to the solves
    # the default preconditioners run there too
    for kind in (1, 3):
        g.set_preconditioner(kind)
        assert g.step() > 0
