"""Pins the CPU oracle (oracle/xpic_oracle.cpp) against the reference's own golden files.

tests/golden/ holds DATA files copied from the reference's tests/<suite>/expected/<test>/:
the `temporal/*.txt` tables (what its `compare_temporal`, tests/common.h:30-90, diffs at
PETSC_SMALL = 1e-10 on values printed as `{: .6e}`) and the float32 field dumps.  Test set-ups
(r0, v0, dt, fields, grid, species) are those of tests/boris_push/boris_push_ex{1..6}.cpp and
tests/{basic,ecsim,ecsimcorr}/*_ex1.cpp.
"""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PETSC_SMALL = 1e-10


def fmt(a):
    """Round the way the reference prints: `{: .6e}`."""
    return np.array([float("% .6e" % v) for v in np.ravel(a)]).reshape(np.shape(a))


BORIS = []
for ex in range(1, 7):
    for f in sorted(glob.glob(os.path.join(GOLD, f"boris_push_ex{ex}", "*.txt"))):
        BORIS.append((ex, os.path.basename(f)[:-4]))


def test_boris_table_count():
    # tests/boris_push/CMakeLists.txt:13-39: ex1 x 13 ids + ex2,ex3 x 4 B ids + ex4,ex5,ex6 x 4 EB ids
    assert len(BORIS) == 13 + 2 * 4 + 3 * 4


@pytest.mark.parametrize("ex,sid", BORIS)
def test_boris_trajectory_tables(oracle, ex, sid):
    """K1/K2 (a2, a3): BorisPush::update_r / update_vEB / update_v{M,B,C1,C2} reproduce all 37 tables."""
    _, gold = oracle.read_table(os.path.join(GOLD, f"boris_push_ex{ex}", sid + ".txt"))
    rows = oracle.boris_trajectory(ex, sid)
    mine = fmt(rows)
    assert mine.shape == gold.shape
    if (ex, sid) == (6, "EB1B"):
        # dt = 2.1*pi makes this trajectory chaotic: after row 99 (t ~ 650/w_pe) last-bit differences
        # of the reference's own -O3 -march=native (FMA-contracted) build are amplified to 1e-6..1e-5.
        # The first 99 rows are exact; the tail is only checked loosely.
        assert np.abs(mine[:99] - gold[:99]).max() < PETSC_SMALL
        assert np.abs(mine[99:] - gold[99:]).max() < 1e-4
    else:
        assert np.abs(mine - gold).max() < PETSC_SMALL


def _integration(oracle, scheme, d, dt):
    oracle.lib().orc_reset_rng()
    s = oracle.OracleSim(scheme, (10, 10, 10), (d, d, d), dt)
    e = s.add_sort(100, 1.0, -1.0, 1.0, (0.1, 0.1, 0.1))
    s.load_maxwell_box(e, True)
    return s


def _energy_row(en):
    return np.array([en[0], en[1], en[4], en[2], en[3], en[5]])


def test_basic_ex1_full_tables(oracle):
    """tests/basic/basic_ex1.cpp: all 100 steps of energy.txt / energy_conservation.txt to every printed
    digit (no linear solve on this path, so nothing but the algorithm enters), plus the float32 E/B dumps."""
    s = _integration(oracle, "basic", 0.05, 0.025)
    # (box volume)*Np/(dx dy dz) truncates to 99999 in double arithmetic, particles_builder.cpp:17,26
    assert s.count(0) == 99999
    _, gold = oracle.read_table(os.path.join(GOLD, "basic_ex1", "energy.txt"))
    _, goldc = oracle.read_table(os.path.join(GOLD, "basic_ex1", "energy_conservation.txt"))
    _, goldq = oracle.read_table(os.path.join(GOLD, "basic_ex1", "charge_conservation.txt"))
    s.charge_collect()
    s.charge_columns()
    prev = s.energy()
    assert np.abs(fmt(_energy_row(prev)) - gold[0, 1:]).max() < PETSC_SMALL
    dump0 = np.fromfile(os.path.join(GOLD, "basic_ex1", "density_000.f32"), dtype=np.float32)
    assert np.array_equal(s.moment_density(0).astype(np.float32).ravel(), dump0)  # initial load: bit-equal
    for t in range(1, 101):
        assert s.step() == 0
        en = s.energy()
        assert np.abs(fmt(_energy_row(en)) - gold[t, 1:]).max() < PETSC_SMALL, t
        cons = np.array([en[0] - prev[0], en[1] - prev[1], en[4] - prev[4]])
        cons = np.append(cons, cons.sum())
        assert np.abs(fmt(cons) - goldc[t, 1:]).max() < PETSC_SMALL, t
        prev = en
        # continuity residual is pure round-off (order of atomic adds): same magnitude as the reference's
        q = s.charge_columns()
        assert q[0] < 4 * goldq[1:, 1].max() and q[1] < 4 * goldq[1:, 2].max()
        if t in (50, 100):
            for name in ("E", "B"):
                dump = np.fromfile(os.path.join(GOLD, "basic_ex1", f"{name}_{t:03d}.f32"), dtype=np.float32)
                mine = s.get_field(name).astype(np.float32).ravel()
                assert np.abs(mine - dump).max() <= 2e-6 * np.abs(dump).max(), (name, t)
            # electrons/density/<t>: DistributionMoment "density", float32 [z][y][x]
            dump = np.fromfile(os.path.join(GOLD, "basic_ex1", f"density_{t:03d}.f32"), dtype=np.float32)
            mine = s.moment_density(0).astype(np.float32).ravel()
            assert np.abs(mine - dump).max() <= 2e-6 * np.abs(dump).max(), ("density", t)


# ---- ecsim_ex1 / ecsimcorr_ex1: all 100 golden rows and the t = 50, 100 dumps -----------------------------------
# What the comparison can support (measured here, oracle at the reference's own tolerances and at rtol = 1e-13 alike):
#   * row 0 (loader) and row 1 (one whole step incl. the solve) agree to every printed digit;
#   * from row 2 on the reference's own KSP stop (atol = 1e-7 absolute on a rhs of norm ~3e-2 -> ~1e-5 relative in
#     the field energies; PETSc GMRES + ILU(0)) makes its table differ from ANY exact solve by
#     2e-5 .. 2e-4 in wE / wB over the first 10 rows, and the thermal-noise fields amplify that difference by ~2e3
#     over 100 steps (4e-2 in wE at row 100).  wK, dominated by the thermal energy, stays within 1.2e-4.
#   * the float32 E dumps therefore agree to 7 % (t = 50) and 32 % (t = 100); density to 0.9 % / 3.4 %;
#   * the golden B dumps of these two tests have the OPPOSITE SIGN of what the present reference code's Faraday update
#     (ecsim/simulation.cpp:247-248 with rotE = -dt rot(+), :553) produces -- basic_ex1's dumps, written through the
#     same Rotor, agree in sign with it -- so B is compared up to that sign (5.7 % at t = 50).
# The tight long-run check of the HIP path is against this oracle at rtol = 1e-13 (tests/test_gpu_host.py).
EARLY = dict(wE=5e-4, wB=5e-4, wK=1e-6)     # rows 2..10
LATE = dict(wE=6e-2, wB=2.5e-2, wK=2e-4)    # rows 11..100
DUMPS = {50: dict(E=0.10, B=0.08, density=0.012), 100: dict(E=0.40, B=0.40, density=0.045)}


def _check_energy_row(t, row, gold_row):
    wE, wB, wK = row[0], row[1], row[2]
    tol = EARLY if t <= 10 else LATE
    assert abs(wE - gold_row[0]) <= tol["wE"] * abs(gold_row[0]), (t, "wE")
    assert abs(wB - gold_row[1]) <= tol["wB"] * abs(gold_row[1]), (t, "wB")
    assert abs(wK - gold_row[2]) <= tol["wK"] * abs(gold_row[2]), (t, "wK")
    # sE, sB, sK follow the same energies
    assert np.allclose(row[3:], gold_row[3:], rtol=tol["wE"], atol=0), (t, "s*")


def _check_dumps(s, name, t):
    lim = DUMPS[t]
    for f in ("E", "B"):
        dump = np.fromfile(os.path.join(GOLD, name, f"{f}_{t:03d}.f32"), dtype=np.float32).astype(np.float64)
        mine = s.get_field(f).astype(np.float32).ravel().astype(np.float64)
        sign = -1.0 if f == "B" else 1.0  # see the note above
        assert np.linalg.norm(mine - sign * dump) <= lim[f] * np.linalg.norm(dump), (f, t)
    dump = np.fromfile(os.path.join(GOLD, name, f"density_{t:03d}.f32"), dtype=np.float32).astype(np.float64)
    mine = s.moment_density(0).astype(np.float32).ravel().astype(np.float64)
    assert np.abs(mine - dump).max() <= lim["density"] * np.abs(dump).max(), ("density", t)


def test_ecsim_ex1_tables(oracle):
    """tests/ecsim/ecsim_ex1.cpp, all 100 steps.  Row 0 pins the RNG-exact loader (wK, sK); row 1 pins the whole ECSIM
    step (a9, a10, a12, a14, a16-a20) to every printed digit; later rows and the dumps within the bounds above."""
    s = _integration(oracle, "ecsim", 0.5, 1.5)
    assert s.count(0) == 100000
    _, gold = oracle.read_table(os.path.join(GOLD, "ecsim_ex1", "energy.txt"))
    _, goldc = oracle.read_table(os.path.join(GOLD, "ecsim_ex1", "energy_conservation.txt"))
    assert gold.shape[0] == 101
    prev = s.energy()
    assert np.abs(fmt(_energy_row(prev)) - gold[0, 1:]).max() < PETSC_SMALL
    dump0 = np.fromfile(os.path.join(GOLD, "ecsim_ex1", "density_000.f32"), dtype=np.float32)
    assert np.array_equal(s.moment_density(0).astype(np.float32).ravel(), dump0)
    for t in range(1, 101):
        assert s.step() > 0
        en = s.energy()
        row = _energy_row(en)
        cons = np.array([en[0] - prev[0], en[1] - prev[1], en[4] - prev[4]])
        prev = en
        if t == 1:
            assert np.abs(fmt(row) - gold[t, 1:]).max() < PETSC_SMALL
            assert np.abs(fmt(cons) - goldc[t, 1:4]).max() < PETSC_SMALL
        else:
            _check_energy_row(t, row, gold[t, 1:])
        # the scheme conserves energy exactly: dE+dB+dK at round-off, like the golden column (~1e-13)
        assert abs(cons.sum()) < 5e-12
        if t in DUMPS:
            _check_dumps(s, "ecsim_ex1", t)


def test_ecsimcorr_ex1_tables(oracle):
    """tests/ecsimcorr/ecsimcorr_ex1.cpp, all 100 steps: row 1 to every printed digit (a13 + second solve on matM)."""
    dt = 1.5
    s = _integration(oracle, "ecsimcorr", 0.5, dt)
    _, gold = oracle.read_table(os.path.join(GOLD, "ecsimcorr_ex1", "energy.txt"))
    _, goldc = oracle.read_table(os.path.join(GOLD, "ecsimcorr_ex1", "energy_conservation.txt"))
    _, goldq = oracle.read_table(os.path.join(GOLD, "ecsimcorr_ex1", "charge_conservation.txt"))
    assert gold.shape[0] == 101
    s.charge_collect()
    s.charge_columns()
    prev = s.energy()
    assert np.abs(fmt(_energy_row(prev)) - gold[0, 1:]).max() < PETSC_SMALL
    for t in range(1, 101):
        assert s.step() > 0
        en = s.energy()
        row = _energy_row(en)
        sc = s.ecsimcorr_scalars(0)
        cwd = sc["lambda_dK"]
        pwd = sc["pred_dK"] - dt * sc["pred_w"]
        ldk = sc["corr_dK"] - dt * sc["corr_w"]
        if t == 1:
            assert np.abs(fmt(row) - gold[t, 1:]).max() < PETSC_SMALL
            assert abs(cwd - goldc[t, 4]) < 2e-10  # last printed digit: KSP stop criterion
        else:
            _check_energy_row(t, row, gold[t, 1:])
            if t <= 10:
                assert abs(cwd - goldc[t, 4]) < 1e-3 * abs(goldc[t, 4])
        assert abs(pwd) < 1e-14 and abs(ldk) < 1e-14  # golden: ~1e-16
        # Esirkepov continuity: N1/N2 norms at the golden's round-off level (~8e-13 / 3.5e-14)
        q = s.charge_columns()
        assert q[0] < 4 * goldq[1:, 1].max() and q[1] < 4 * goldq[1:, 2].max()
        prev = en
        if t in DUMPS:
            _check_dumps(s, "ecsimcorr_ex1", t)


def test_basic_ex1_momentum_table(oracle):
    """MomentumConservation is always on (src/interfaces/simulation.cpp:55-56): all 100 rows of
    tests/basic/expected/basic_ex1/temporal/momentum_conservation.txt.  P, QE and the defect columns to the printed
    digits; fP of row 1 (5e-13: P did not change, E = 0 at t = 0) is pure round-off and only bounded."""
    s = _integration(oracle, "basic", 0.05, 0.025)
    dt = 0.025
    _, gold = oracle.read_table(os.path.join(GOLD, "basic_ex1", "momentum_conservation.txt"))
    assert gold.shape == (101, 10)
    p0 = s.momentum()[0]
    assert np.abs(fmt(p0) - gold[0, 1:7]).max() < PETSC_SMALL
    for t in range(1, 101):
        assert s.step() == 0
        p1 = s.momentum()[0]
        assert np.abs(fmt(p1) - gold[t, 1:7]).max() < PETSC_SMALL, t
        err = (p1[:3] - p0[:3]) / dt - p1[3:]
        n2 = np.sqrt((err * err).sum())
        den = np.sqrt(((p1[:3] + p0[:3]) ** 2).sum())
        freq = np.sqrt(((p1[:3] - p0[:3]) ** 2).sum()) / den / (0.5 * dt) if den > 1e-10 else 0.0
        # (p1 - p0) / dt cancels 3-4 digits: the defect is good to ~1e-9, its printed value to 2 units of the 7th digit
        assert abs(n2 - gold[t, 7]) < 2e-9 and abs(n2 - gold[t, 9]) < 2e-9, t
        if t == 1:
            assert freq < 1e-10
        else:
            assert abs(float(fmt(freq)) - gold[t, 8]) <= 2e-6 * abs(gold[t, 8]), t  # 2 units of the 7th printed digit
        p0 = p1
