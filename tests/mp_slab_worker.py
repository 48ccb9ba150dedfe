"""Worker of tests/test_gpu_slabs.py: `nranks` processes share ONE GPU, each owns a z-slab, the exchange goes
through gloo (xpic_comm_init_callbacks).  Every rank runs the same seeded problem; rank 0 also runs it on a
single-slab context AND on the CPU oracle (whole box) and checks that the decomposed run reproduces both: fields,
particle totals and the per-cell occupancy of the whole box (update_cells_mpi, src/interfaces/particles.cpp:118-248).
usage: mp_slab_worker.py <scheme> [planes per slab]
environment: XPIC_SLAB_PEER=1 -- the matL ghost rows travel by hipMemcpyAsync into the neighbours' IPC-mapped buffers;
XPIC_SLAB_GATHER_WINDOW=<slots> -- the gathering assembly reaches only that far by 32-bit offsets (the far
arm and the receive-buffer arm of its gather then meet the oracle); XPIC_SLAB_CONFINE=1 -- the LAST species lives in the
middle of slab 0 only and is cold, so the other slabs hold no particle of it (the matL ghost-row exchange must still be
posted at the same place of every rank's message sequence); XPIC_SLAB_CLUMP=1 -- 110 slow particles of the first
species sit in ONE cell of the first plane of slab 1 and 60 more fly towards it from two planes below (slab 0), one plane per
step: at the second step they arrive through the migration and the cell holds more than a bucket of the deferred scatter
(128), so that slab falls back to the index built from the keys -- which the pre-binning second push did not write: they
are rebuilt -- while the other slabs keep their buckets"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xpic_amd as X  # noqa: E402
from xpic_amd.parallel import GlooRing  # noqa: E402


SORTS = [(8, 1.0, -1.0, 1.0), (8, 1.0, 1.0, 16.0)]


def problem(scheme, n, d, seed):
    """The seeded inputs: particles per sort and the E, B, B0 arrays of the whole box."""
    rng = np.random.default_rng(seed)
    vth = 0.25 if scheme == "ecsim" else 0.1  # Esirkepov moves must stay below one cell (dz = 0.25, dt = 0.2)
    N = n[0] * n[1] * n[2]
    L = np.array(n) * np.array(d)
    parts = []
    for i, _ in enumerate(SORTS):
        pts = np.empty((8 * N, 6))
        pts[:, :3] = rng.random((8 * N, 3)) * L
        pts[:, 3:] = rng.normal(0, vth, (8 * N, 3))
        if CLUMP and i == 0:
            cell = np.array([3, 2, CLUMP]) * np.array(d)  # (x, y, first plane of slab 1)
            pts[:170, :3] = cell + (0.3 + 0.4 * rng.random((170, 3))) * np.array(d)
            pts[:170, 3:] *= 0.02  # (slow: they are still there after the steps)
            pts[110:170, 2] -= 2 * d[2]  # two planes below: slab 0 ...
            pts[110:170, 5] += d[2] / 0.8  # ... and one plane per step of ecsim's dt = 0.8 upwards
        if CONFINE and i == len(SORTS) - 1:
            pts = pts[: N]
            pts[:, 2] = (0.3 + 0.4 * rng.random(N)) * CONFINE * d[2]  # CONFINE = planes of slab 0
            pts[:, 3:] *= 0.02
        parts.append(pts)
    shape = (n[2], n[1], n[0], 3)
    E = rng.normal(0, 0.02, shape)
    B = rng.normal(0, 0.02, shape) + np.array([0.0, 0.1, 0.3])
    B0 = np.zeros(shape) + np.array([0.0, 0.1, 0.3])
    return parts, E, B, B0


CONFINE = 0  # planes of slab 0 (set by main() from XPIC_SLAB_CONFINE)
CLUMP = 0    # planes per slab (set by main() from XPIC_SLAB_CLUMP)
RCCL = os.environ.get("XPIC_SLAB_TRANSPORT") == "rccl"  # one GPU per rank, RCCL over xGMI (needs >= nranks devices)


def build(scheme, n, d, dt, rank, nranks, seed):
    parts, E, B, B0 = problem(scheme, n, d, seed)
    dev = int(os.environ.get("LOCAL_RANK", "0")) if RCCL and nranks > 1 else 0
    ctx = X.Context(scheme, n, d, dt, device=dev, rank=rank, nranks=nranks)
    N = n[0] * n[1] * n[2]
    for (Np, dens, q, m), pts in zip(SORTS, parts):
        s = ctx.add_sort(Np, dens, q, m, capacity=3 * 8 * N)
        ctx.add_particles(s, pts)  # add_particle keeps what lies in the local slab
    z0, nzl = ctx.z0, ctx.nzl
    for fid, F in ((X.E, E), (X.B, B), (X.B0, B0)):
        ctx.set_field(fid, F[z0:z0 + nzl])
    ctx.set_tolerances(1e-12, 1e-50, 400)
    if os.environ.get("XPIC_SLAB_GATHER_WINDOW"):
        ctx.debug_set(X.DEBUG_GATHER_WINDOW, int(os.environ["XPIC_SLAB_GATHER_WINDOW"]))
    return ctx


def build_oracle(scheme, n, d, dt, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    oracle_lib.build()
    parts, E, B, B0 = problem(scheme, n, d, seed)
    o = oracle_lib.OracleSim(scheme, n, d, dt)
    for (Np, dens, q, m), pts in zip(SORTS, parts):
        so = o.add_sort(Np, dens, q, m)
        assert o.add_particles(so, pts) == pts.shape[0]
    for name, F in (("E", E), ("B", B), ("B0", B0)):
        o.set_field(name, F)
    o.set_tolerances(1e-12, 1e-50, 400)
    return o


def gather_field(ctx, fid, nranks):
    loc = ctx.get_field(fid)
    parts = [None] * nranks
    dist.all_gather_object(parts, loc)
    return np.concatenate(parts, axis=0)


def main():
    scheme = sys.argv[1]
    nzl = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    global CONFINE, CLUMP
    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    if os.environ.get("XPIC_SLAB_CONFINE") == "1":
        CONFINE = nzl
    if os.environ.get("XPIC_SLAB_CLUMP") == "1":
        CLUMP = nzl
    n, d = (12, 10, nzl * nranks), (0.5, 0.4, 0.25)
    dt = 0.2 if scheme != "ecsim" else 0.8
    ctx = build(scheme, n, d, dt, rank, nranks, seed=42)
    if RCCL:
        from xpic_amd.parallel import init_rccl

        init_rccl(ctx, device=int(os.environ.get("LOCAL_RANK", "0")))
        ctx.set_overlap(int(os.environ.get("XPIC_SLAB_OVERLAP", "0")))  # 0 blocking, 1 operator halos, 3 + matL ghost rows
    else:
        GlooRing().attach(ctx)
    if os.environ.get("XPIC_SLAB_PEER") == "1" and scheme != "basic":
        # the matL ghost rows by hipMemcpyAsync into the neighbours' IPC-mapped receive buffers (the copy-engine path)
        from xpic_amd.parallel import map_peers

        map_peers(ctx)
        ctx.set_overlap(4 | int(os.environ.get("XPIC_SLAB_OVERLAP", "0")))
        ctx.profile_enable(True)
    counts0 = [ctx.count(s) for s in range(2)]
    if CONFINE:
        assert (counts0[1] > 0) == (rank == 0), counts0  # the last species lives on slab 0 alone
    nsteps = 3
    if CLUMP:
        ctx.profile_enable(True)
    its = [ctx.step() for _ in range(nsteps)]
    if CLUMP and scheme == "ecsim":
        # the clump's slab went through the index from the second step on (its keys rebuilt once, after the key-less
        # pre-binning that the arrivals overflowed), the others kept their buckets; every slab's assembly gathered
        idx, rebuilt, gathered = (ctx.profile_get(k)[0] for k in ("index", "rebuild_keys", "fill_gather"))
        assert gathered == 2 * nsteps, gathered
        assert (idx, rebuilt) == ((nsteps - 1, 1) if rank == 1 else (0, 0)), (rank, idx, rebuilt)
        ctx.profile_enable(False)
    if os.environ.get("XPIC_SLAB_PEER") == "1" and scheme != "basic":
        assert ctx.profile_get("peer_copies")[0] == 4 * nsteps, ctx.profile_get("peer_copies")  # 3 planes up + 1 down per step
        ctx.profile_enable(False)
    en = ctx.energy()
    fields = {name: gather_field(ctx, fid, nranks) for name, fid in (("E", X.E), ("B", X.B))}
    counts = [ctx.count(s) for s in range(2)]
    allc = [None] * nranks
    dist.all_gather_object(allc, (counts0, counts))
    # per-cell occupancy of the whole box: local cell + the cells below this slab
    occ = [np.bincount(ctx.particles(s)[1].astype(np.int64) + ctx.z0 * n[0] * n[1], minlength=n[0] * n[1] * n[2])
           for s in range(2)]
    allocc = [None] * nranks
    dist.all_gather_object(allocc, occ)
    ok = True
    if rank == 0:
        ref = build(scheme, n, d, dt, 0, 1, seed=42)
        rits = [ref.step() for _ in range(nsteps)]
        ren = ref.energy()
        tot0 = [sum(c[0][s] for c in allc) for s in range(2)]
        tot1 = [sum(c[1][s] for c in allc) for s in range(2)]
        print("particles before/after", tot0, tot1, "single-slab", [ref.count(s) for s in range(2)], flush=True)
        ok &= tot1 == [ref.count(s) for s in range(2)]
        ok &= any(c[1] != c[0] for c in allc)  # particles did migrate between the slabs
        if CONFINE:
            ok &= all(c[1][1] == 0 for c in allc[1:])  # ... but the confined species stayed where it was
        for name, fid in (("E", X.E), ("B", X.B)):
            a = ref.get_field(fid)
            err = np.abs(a - fields[name]).max() / np.abs(a).max()
            print(name, "rel err vs single slab", err, flush=True)
            ok &= err < 1e-8
        print("energy", en, ren, "its", its, rits, flush=True)
        ok &= np.allclose(en, ren, rtol=1e-9, atol=1e-15)
        if scheme != "basic":
            ok &= all(abs(a - b) <= 2 for a, b in zip(its, rits))
        # ---- the same problem on the CPU oracle, whole box
        o = build_oracle(scheme, n, d, dt, seed=42)
        for _ in range(nsteps):
            assert o.step() >= 0
        for name in ("E", "B"):
            a = o.get_field(name)
            err = np.abs(a - fields[name]).max() / np.abs(a).max()
            print(name, "rel err vs oracle", err, flush=True)
            ok &= err < 1e-8
        for s in range(2):
            ok &= o.count(s) == tot1[s]
            oo = np.bincount(o.particles(s)[1].astype(np.int64), minlength=n[0] * n[1] * n[2])
            mine = sum(r[s] for r in allocc)
            same = np.array_equal(oo, mine)
            print("sort", s, "per-cell occupancy equals the oracle's:", same, flush=True)
            ok &= same
        ok &= np.allclose(en, o.energy(), rtol=1e-9, atol=1e-15)
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    if not flag[0]:
        sys.exit(1)
    print(f"rank {rank}/{nranks} ok", flush=True)


if __name__ == "__main__":
    main()
