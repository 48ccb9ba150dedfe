"""CPU-only driver of the oracle entry points that the `-m gpu` tests call (run by tests/test_oracle_sanitized.py in a
child process, against an AddressSanitizer + UBSan build of oracle/xpic_oracle.cpp named by XPIC_ORACLE_SO).

It mirrors the call sequences of tests/test_gpu_ecsim.py, test_gpu_schemes.py, test_gpu_edges.py and
test_eccapfim_kernels.py (make_pair: random particles + random fields, then the phase functions, the solves and whole
steps) at several OpenMP team sizes, and also the MISUSE sequences a half-written test can produce (a solve or an apply
on a simulation whose matL was never assembled).  Prints "SAN-DRIVER-OK" at the end; the sanitizer aborts the process
with a report otherwise.
"""
import sys

import numpy as np

import oracle_lib as O

GRID = ((12, 10, 8), (0.5, 0.4, 0.25))


def make(scheme, n, d, dt, sorts, ppc=6, vth=0.05, B0=(0.0, 0.0, 0.0), seed=0):
    rng = np.random.default_rng(seed)
    o = O.OracleSim(scheme, n, d, dt)
    N = n[0] * n[1] * n[2]
    for (Np, dens, q, m) in sorts:
        so = o.add_sort(Np, dens, q, m)
        pts = np.empty((ppc * N, 6))
        pts[:, :3] = rng.random((ppc * N, 3)) * (np.array(n) * np.array(d))
        pts[:, 3:] = rng.normal(0, vth, (ppc * N, 3))
        assert o.add_particles(so, pts) == ppc * N
    for name in ("E", "B"):
        F = rng.normal(0, 0.05, o.fshape())
        if name == "B":
            F += np.array(B0)
        o.set_field(name, F)
    o.set_field("B0", np.zeros(o.fshape()) + np.array(B0))
    return o


def ecsim_sequences(threads):
    L = O.lib()
    L.orc_set_threads(threads)
    n, d = GRID
    # misuse first: applies and solves before any assembly (matL is the zero matrix then, not an empty array)
    o = make("ecsim", n, d, 0.7, [])
    F = o.get_field("E")
    assert np.abs(o.matL_apply(F)).max() == 0.0
    assert np.abs(o.matL()).max() == 0.0
    x, its, rn = o.solve(0, F, 1e-9, 1e-50, 50)  # matA = matM alone
    assert its > 0 and np.isfinite(x).all()
    # test_fill_current_and_matL, test_solve_matches_oracle, test_second_push_matches_oracle
    o = make("ecsim", n, d, 1.5, [(8, 1.0, -1.0, 1.0), (3, 0.5, 1.0, 20.0)], B0=(0.0, 0.0, 0.5))
    L.orc_ecsim_first_push(o.h, 0)
    L.orc_update_cells(o.h, 0)
    L.orc_ecsim_fill_current(o.h)
    assert np.isfinite(o.matL()).all() and np.abs(o.sort_current(0, "currI")).max() > 0
    rhs = o.get_field("E")
    for op in (0, 1, 2):
        x, its, rn = o.solve(op, rhs, 1e-9, 1e-50, 300)
        assert its > 0 and np.isfinite(x).all(), (op, its)
    x, its, rn = o.solve(0, rhs, 1e-30, 1e-50, 3)  # runs out of iterations: negative count, no fault
    assert its < 0
    L.orc_ecsim_second_push(o.h, 0)
    o.set_tolerances(1e-10, 1e-50, 300)
    for _ in range(2):
        assert o.step() >= 0
    assert np.isfinite(o.energy()).all()
    # tiny extents like the reference's default config (test_tiny_extents...)
    for nn in ((2, 2, 32), (3, 2, 5), (5, 3, 2)):
        o = make("ecsim", nn, (0.5, 0.5, 0.5), 1.0, [(4, 1.0, -1.0, 1.0)], ppc=4)
        assert o.step() >= 0


def scheme_sequences(threads):
    L = O.lib()
    L.orc_set_threads(threads)
    n, d = GRID
    o = make("ecsimcorr", n, d, 0.7, [(6, 1.0, -1.0, 1.0), (3, 1.0, 1.0, 50.0)], B0=(0.0, 0.1, 0.5), vth=0.1)
    assert L.orc_ecsimcorr_first_push(o.h, 0) == 0
    L.orc_ecsim_fill_current(o.h)
    assert L.orc_ecsimcorr_second_push(o.h, 0) == 0
    L.orc_ecsimcorr_final_update(o.h, 0)
    assert np.isfinite(L.orc_calculate_energy(o.h, 0))
    o.ecsimcorr_scalars(0)
    o.momentum()
    for _ in range(2):
        assert o.step() >= 0
    o.charge_collect()
    o.charge_columns()
    o.charge_density(0)
    o.moment_density(1)
    o = make("basic", n, d, 0.1, [(6, 1.0, -1.0, 1.0), (3, 1.0, 1.0, 50.0)], vth=0.1)
    assert L.orc_basic_push(o.h) == 0
    for _ in range(2):
        assert o.step() >= 0
    o.momentum()
    # the RNG-exact loader of the golden runs
    L.orc_reset_rng()
    o = O.OracleSim("basic", (8, 8, 8), (0.5, 0.5, 0.5), 0.1)
    o.add_sort(8, 1.0, -1.0, 1.0, T=(0.1, 0.1, 0.1))
    assert o.load_maxwell_box(0) > 0
    o.clear(0)
    assert o.count(0) == 0


def eccapfim_sequences(threads):
    """test_device_kernels_match_oracle: segments from cell_traversal, then interpolate / decompose on them"""
    L = O.lib()
    L.orc_set_threads(threads)
    rng = np.random.default_rng(3)
    n, d = (9, 8, 7), (0.5, 0.4, 0.3)
    o = make("basic", n, d, 0.7, [])
    Lb = np.array(n) * np.array(d)
    r0 = rng.random((200, 3)) * Lb * 0.6 + Lb * 0.2
    rn = r0 + (rng.random((200, 3)) * 2 - 1) * 1.4 * np.array(d)
    se, ss = [], []
    for q in range(200):
        pts, cnt = o.cell_traversal(rn[q], r0[q], 16)
        assert cnt == len(pts) >= 2
        for k in range(1, cnt):
            se.append(pts[k])
            ss.append(pts[k - 1])
    se, ss = np.array(se), np.array(ss)
    o.implicit_esirkepov_interpolate(se, ss)
    o.set_field("J", np.zeros(o.fshape()))
    o.implicit_esirkepov_decompose(rng.random(len(se)), rng.normal(0, 1, (len(se), 3)), se, ss, "J")
    out = np.zeros((n[2], n[1], n[0]))
    L.orc_div_neg(o.h, O._dp(np.ascontiguousarray(o.get_field("J"))), O._dp(out))


if __name__ == "__main__":
    for threads in (1, 3, 16):
        ecsim_sequences(threads)
        scheme_sequences(threads)
        eccapfim_sequences(threads)
    print("SAN-DRIVER-OK")
    sys.stdout.flush()
