"""usage: precond_spectrum.py [grid] [ppc]   (CPU only: the oracle assembles matL, numpy / scipy do the rest)
Spectral experiment behind precond.hip: how many GMRES iterations does matA need with P = matM^-1, P = (matM + Lbar)^-1 (Lbar = translation
average of matL) and Chebyshev approximations of the latter?  Small periodic grid, bench parameters (dx .5, dt 1, 64 ppc)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import xpic_amd as X
import scipy.sparse as sp
import scipy.sparse.linalg as spl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dx, dt, b0, vth = 0.5, 1.0, 0.2, 0.014
o = oracle_lib.OracleSim("ecsim", (n, n, n), (dx,) * 3, dt)
s = o.add_sort(ppc, 1.0, -1.0, 1.0)
rng = np.random.default_rng(1)
npart = ppc * n ** 3
pts = np.empty((npart, 6))
pts[:, :3] = rng.random((npart, 3)) * (n * dx)
pts[:, 3:] = rng.normal(0, vth, (npart, 3))
o.add_particles(s, pts)
B = np.zeros(o.fshape()); B[..., 2] = b0
o.set_field("B", B); o.set_field("B0", B)
oracle_lib.lib().orc_ecsim_fill_current(o.h)
Lm = o.matL()   # [node][c1][k]
print("matL", Lm.shape)
N = n ** 3
Lm = Lm.reshape(n, n, n, 3, -1)  # z y x c1 k
K = Lm.shape[-1]
dec = [[X.lstencil_decode(c1, k) for k in range(K)] for c1 in range(3)]
idx = np.arange(N).reshape(n, n, n)
def build(Lcoef):
    rows, cols, vals = [], [], []
    zz, yy, xx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    for c1 in range(3):
        for k in range(K):
            c2, d = dec[c1][k]
            col = idx[(zz + d[2]) % n, (yy + d[1]) % n, (xx + d[0]) % n] * 3 + c2
            rows.append((idx * 3 + c1).ravel()); cols.append(col.ravel()); vals.append(Lcoef[..., c1, k].ravel())
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * N, 3 * N))
Lsp = build(Lm)
Lbar = np.broadcast_to(Lm.mean(axis=(0, 1, 2)), Lm.shape)
Lbsp = build(Lbar)
# matM as a matrix via the oracle apply on unit vectors (3N columns) -- use linear operator instead
def matM(v):
    return o.matM(v.reshape(o.fshape())).ravel()
# field layout of oracle: [z][y][x][c] -> index (node*3 + c): matches idx*3+c
Mcols = []
I = np.eye(3 * N)
t0 = time.time()
Mden = np.empty((3 * N, 3 * N))
for j in range(3 * N):
    Mden[:, j] = matM(I[j])
print("matM dense built", time.time() - t0)
A = Mden + Lsp.toarray()
Ab = Mden + Lbsp.toarray()
rhs = rng.normal(0, 1, 3 * N)
def gmres_its(P, label):
    its = [0]
    def cb(rk): its[0] += 1
    x, info = spl.gmres(spl.aslinearoperator(A @ P) if P is not None else spl.aslinearoperator(A), rhs, rtol=1e-7, atol=0, restart=30, maxiter=20, callback=cb, callback_type="pr_norm")
    print("%-40s iterations %d (info %d)" % (label, its[0], info))
gmres_its(None, "no preconditioner")
Minv = np.linalg.inv(Mden)
gmres_its(Minv, "P = matM^-1 (exact)")
Abinv = np.linalg.inv(Ab)
gmres_its(Abinv, "P = (matM + Lbar)^-1 (exact)")
ev = np.linalg.eigvals(A @ Abinv)
print("spectrum of A (M+Lbar)^-1: real [%.4f, %.4f], |imag| max %.4f" % (ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
ev = np.linalg.eigvals(A @ Minv)
print("spectrum of A M^-1: real [%.4f, %.4f], |imag| max %.4f" % (ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
# Chebyshev polynomial in (M + Lbar), spectral interval from eigenvalues of its symmetric part
evb = np.linalg.eigvals(Ab)
print("spectrum of M+Lbar: real [%.4f, %.4f] imag %.4f" % (evb.real.min(), evb.real.max(), np.abs(evb.imag).max()))
a, b = evb.real.min(), evb.real.max()
def cheb(deg):
    theta, delta = 0.5 * (b + a), 0.5 * (b - a)
    sigma1 = theta / delta
    rho = 1.0 / sigma1
    # matrix polynomial: apply iteration to identity columns
    R = np.eye(3 * N)
    Z = R / theta
    D = Z.copy()
    for i in range(1, deg):
        rho_new = 1.0 / (2 * sigma1 - rho)
        D = rho_new * rho * D + (2 * rho_new / delta) * (R - Ab @ Z)
        Z = Z + D
        rho = rho_new
    return Z
for deg in (6, 8, 10, 12):
    gmres_its(cheb(deg), "P = Chebyshev_%d(matM + Lbar)" % deg)

# ---- round 4 (review item 5): surrogates that see the LOCAL density.  Lbar is exact for a uniform plasma; what is left for
# GMRES is the particle noise of matL, about half of which is count noise (Poisson(ppc) particles per cell).
#   (a) rows of Lbar scaled by the local density seen in the row's own diagonal entry:  Abar_a = matM + diag(r) Lbar,
#       r[node, c1] = matL[node][c1][diag] / Lbar[c1][diag];
#   (b) the same, and the exact local 3 x 3 same-node block of matL in place of Lbar's centre taps;
#   (c) symmetric scaling  diag(sqrt r) Lbar diag(sqrt r)  (keeps the surrogate's symmetric part positive definite).
# Exact inverses on the small box: if none of them gives 3 iterations at the reference's tolerance, no kernel is built.
kdiag = [[k for k in range(K) if dec[c1][k] == (c1, (0, 0, 0))][0] for c1 in range(3)]
Lb0 = Lm.mean(axis=(0, 1, 2))                                   # [c1][k]
r = np.stack([Lm[..., c1, kdiag[c1]] / Lb0[c1, kdiag[c1]] for c1 in range(3)], axis=-1)   # z y x c1
print("local density ratio r: mean %.4f std %.4f min %.3f max %.3f" % (r.mean(), r.std(), r.min(), r.max()))
La = Lbar * r[..., None]
Aa = Mden + build(La).toarray()
gmres_its(np.linalg.inv(Aa), "P = (matM + diag(r) Lbar)^-1 (exact)")
ev = np.linalg.eigvals(A @ np.linalg.inv(Aa))
print("  spectrum of A Abar_a^-1: real [%.4f, %.4f], |imag| max %.4f" % (ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
Lbl = La.copy()
for c1 in range(3):
    for k in range(K):
        if dec[c1][k][1] == (0, 0, 0):
            Lbl[..., c1, k] = Lm[..., c1, k]
Ab2 = Mden + build(Lbl).toarray()
gmres_its(np.linalg.inv(Ab2), "P = (matM + diag(r) Lbar, local 3x3)^-1")
ev = np.linalg.eigvals(A @ np.linalg.inv(Ab2))
print("  spectrum: real [%.4f, %.4f], |imag| max %.4f" % (ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
sr = np.sqrt(np.maximum(r, 0.0)).reshape(-1)                    # node-major, c1 fastest: the vector layout
Ac = Mden + (sr[:, None] * Lbsp.toarray()) * sr[None, :]
gmres_its(np.linalg.inv(Ac), "P = (matM + sqrt(r) Lbar sqrt(r))^-1 (exact)")
ev = np.linalg.eigvals(A @ np.linalg.inv(Ac))
print("  spectrum: real [%.4f, %.4f], |imag| max %.4f" % (ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
# reference points: what the count noise alone is worth -- the surrogate with the EXACT cell densities (not available to a
# constant stencil) and the ideal of a block-diagonal correction
for tol in (1e-7, 1e-6):
    for (Pm, label) in ((Abinv, "Lbar"), (np.linalg.inv(Ac), "sqrt(r) Lbar sqrt(r)")):
        its = [0]
        def cb(rk): its[0] += 1
        x, info = spl.gmres(spl.aslinearoperator(A @ Pm), rhs, rtol=tol, atol=0, restart=30, maxiter=20, callback=cb, callback_type="pr_norm")
        res = np.linalg.norm(rhs - A @ Pm @ x) / np.linalg.norm(rhs)
        print("rtol %.0e  %-24s iterations %d, true relative residual %.2e" % (tol, label, its[0], res))

# ---- the same with the surrogate's inverse replaced by the fixed Chebyshev polynomial the GPU applies (fp64 here)
def cheb_of(Am, lo, hi, deg):
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma1 = theta / delta
    rho = 1.0 / sigma1
    R = np.eye(3 * N)
    Z = R / theta
    D = Z.copy()
    for i in range(1, deg):
        rho_new = 1.0 / (2 * sigma1 - rho)
        D = rho_new * rho * D + (2 * rho_new / delta) * (R - Am @ Z)
        Z = Z + D
        rho = rho_new
    return Z
rs = np.abs(Lb0).sum(axis=1).max()
hi0 = 2.0 + 2.0 * dt * dt * 3.0 / (dx * dx)
for (Am, label, hi) in ((Aa, "matM + diag(r) Lbar", hi0 + r.max() * rs), (Ac, "matM + sqrt(r) Lbar sqrt(r)", hi0 + r.max() * rs)):
    eva = np.linalg.eigvals(Am)
    print("%s: spectrum real [%.4f, %.4f], |imag| max %.4f; interval used [2, %.3f]" % (label, eva.real.min(), eva.real.max(), np.abs(eva.imag).max(), hi))
    for deg in (8, 10, 12, 14):
        P = cheb_of(Am, 2.0, hi, deg)
        its = [0]
        def cb(rk): its[0] += 1
        x, info = spl.gmres(spl.aslinearoperator(A @ P), rhs, rtol=1e-7, atol=0, restart=30, maxiter=20, callback=cb, callback_type="pr_norm")
        res = np.linalg.norm(rhs - A @ P @ x) / np.linalg.norm(rhs)
        print("  Chebyshev_%-2d  iterations %d, true relative residual %.2e" % (deg, its[0], res))
