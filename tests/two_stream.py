"""The cold two-stream instability: the one physics result BASELINE configs[1] ("3D electrostatic two-stream") can have.

Two equal counter-streaming electron beams (+-v0 along x, each half of the total density n0; the `basic` scheme advances E
by Ampere's law alone, so the uniform mean charge plays the role of a fixed neutralising background).  Cold-fluid
dispersion with w_b^2 = w_pe^2 / 2 per beam:

    1 = w_b^2 [1 / (w - k v0)^2 + 1 / (w + k v0)^2]
    =>  w^2 = k^2 v0^2 + w_b^2 - w_b sqrt(4 k^2 v0^2 + w_b^2)          (the unstable branch; < 0 for k v0 < sqrt(2) w_b)
    =>  gamma(k) = sqrt(-w^2),    gamma_max = w_b / 2 = w_pe / (2 sqrt 2)  at  k v0 = (sqrt 3 / 2) w_b.

In xpic's units (lengths c / w_pe, times 1 / w_pe, densities n0) w_pe = 1 for the total density n = 1.  The reference has no
fixture for this (its JSON surface cannot even express a drifting Maxwellian: src/interfaces/simulation.tpp:24-41 never reads
SortParameters::px) -- **parity unpinned**; what is checked is theory.

The load is a quiet start: every beam's particles sit on a regular lattice, displaced along x by eps sin(k x) with opposite signs for the two beams, so that the seeded mode `m` grows out of a noise
floor many decades below it and its field energy W_E ~ exp(2 gamma t) can be fitted over a long window."""
import numpy as np

V0 = 0.2          # beam speed (c)
NX, NYZ = 64, 6   # cells: quasi-1D box (the 2nd-order shapes of `basic` need >= 6 cells per axis)
DX = 0.125        # c / w_pe: the box is 8 long; mode 4 has k = pi, k v0 = 0.628 = 0.889 w_b (gamma = 0.4997 w_b)
MODE = 4
DT = 0.05
PPC_BEAM = 16     # per beam and cell: BASELINE configs[1]'s two species of 16
EPS = 1e-5        # seed displacement (c / w_pe)


def gamma_theory(k, v0=V0, wpe=1.0):
    wb2 = 0.5 * wpe * wpe
    w2 = k * k * v0 * v0 + wb2 - np.sqrt(wb2) * np.sqrt(4.0 * k * k * v0 * v0 + wb2)
    return np.sqrt(-w2) if w2 < 0 else 0.0


def beams(nx=NX, nyz=NYZ, dx=DX, ppc=PPC_BEAM, v0=V0, mode=MODE, eps=EPS):
    """[(points6 of beam +v0), (points6 of beam -v0)], box (nx, nyz, nyz), spacing, k of the seeded mode"""
    L = nx * dx
    k = 2.0 * np.pi * mode / L
    sub = {16: (4, 2, 2), 8: (2, 2, 2), 32: (4, 4, 2)}[ppc]  # the beam's particles of a cell: a regular sub-lattice
    out = []
    for b, sgn in enumerate((+1.0, -1.0)):
        # (the second beam's lattice is shifted by half a lattice spacing: the two beams do not start on top of each other)
        ax = [(np.arange(m * s_) + 0.5 + 0.5 * b * (a == 0)) * (dx / s_) for a, (m, s_) in enumerate(zip((nx, nyz, nyz), sub))]
        X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
        x0 = X.ravel()
        pts = np.zeros((x0.size, 6))
        pts[:, 0] = np.mod(x0 + sgn * eps * np.sin(k * x0), L)
        pts[:, 1] = Y.ravel()
        pts[:, 2] = Z.ravel()
        pts[:, 3] = sgn * v0
        out.append(pts)
    return out, (nx, nyz, nyz), (dx, dx, dx), k


def fit_growth(t, wE, lo=1e-3, hi=1e-1):
    """gamma from the slope of log W_E(t) / 2 over the window where W_E lies between lo and hi times its maximum (the
    linear phase: well above the seed's transient, below saturation); returns (gamma, points used)"""
    wE = np.asarray(wE)
    i_sat = int(np.argmax(wE))
    sel = [i for i in range(i_sat) if lo * wE[i_sat] <= wE[i] <= hi * wE[i_sat]]
    # (the window is contiguous for a monotone exponential rise; guard against a noisy start)
    sel = [i for i in sel if i >= sel[-1] - (len(sel) - 1)] if sel else sel
    if len(sel) < 8:
        return float("nan"), len(sel)
    slope = np.polyfit(np.asarray(t)[sel], np.log(wE[sel]), 1)[0]
    return 0.5 * slope, len(sel)
