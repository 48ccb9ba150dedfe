#!/usr/bin/env python3
"""bench.py -- xpic hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" is one full timestep of the selected scheme (default: ECSIM = first_push + re-bin, current/mass-matrix
assembly, implicit field solve, second_push, field update) of BASELINE.json's headline configuration: 256^3 cells,
64 particles per cell, one electron species, uniform B0 -- all resident in HBM before the timed region.

One process per GPU.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment STARTS the N ranks
itself (fresh child processes, one per GPU, RCCL over xGMI; the parent never touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` each process is one of the ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate
FP64_PEAK_TF = 78.6     # fp64 vector = fp64 matrix peak of MI355X (measured with v_mfma_f64_16x16x4: 75.5)
FILL_FLOP_PER_PARTICLE = 1200.0  # 576 products + adds of decompose_ecsim_current + the per-particle algebra


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=None, help="cells per axis (default 256; 128 for the side schemes)")
    ap.add_argument("--grid-xyz", type=int, nargs=3, default=None, metavar=("NX", "NY", "NZ"),
                    help="non-cubic box, e.g. 512 512 64 = one GPU's slab of BASELINE configs[4] (overrides --grid)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="wall-clock limit (s) of a --gpus N run started by this launcher: on expiry the ranks are stopped")
    ap.add_argument("--ppc", type=int, default=None, help="particles per cell (default 64; 32 for the side schemes)")
    ap.add_argument("--dx", type=float, default=0.5)
    ap.add_argument("--dt", type=float, default=None, help="default 1.0 (ecsim, ecsimcorr), 0.1 (basic: explicit, CFL)")
    ap.add_argument("--vth", type=float, default=0.014)  # T = 0.1 keV electrons (tests/ecsim/ecsim_ex1.cpp:66-70)
    ap.add_argument("--b0", type=float, default=0.2)
    ap.add_argument("--loader", default="poisson", choices=["poisson", "regular", "gradient", "blob"],
                    help="poisson: positions uniform over the box like CoordinateInBox (Poisson occupancy of the cells); "
                         "regular: exactly ppc particles in every cell; gradient: density falling 4 : 1 along x; blob: 1 %% of "
                         "the particles in a Gaussian clump of sigma = 4 cells at the centre (cells of several hundred particles); "
                         "the same particle total for all of them")
    ap.add_argument("--drift", type=float, default=None,
                    help="basic: beam momentum +-drift (m c) of the two electron species along x (default 0.2: the two-stream "
                         "set-up of BASELINE configs[1]; 0: two thermal species)")
    ap.add_argument("--cpu-grid", type=int, default=64)   # 64^3 x ppc, as SURVEY 8(d) asks: ~20 s of oracle work
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", dest="probe", action="store_false",
                    help="skip the device copy probe (1 GiB device-to-device copies: the achievable copy rate of THIS box, "
                         "SURVEY 8(d); on by default, outside the timed region)")
    ap.add_argument("--plain-gmres", action="store_true", help="unpreconditioned GMRES(30), as the CPU oracle runs")
    ap.add_argument("--fused-rebin", type=int, default=None, choices=[0, 1, 2],
                    help="ecsim: 1 the re-binning's scatter deferred into the assembly's particle loads, 0 scatter first (default: the library's)")
    ap.add_argument("--fill-kernel", type=int, default=None, choices=[0, 1],
                    help="mass-matrix assembly: 1 warp-specialised kernel, 0 the classic 4-wave kernel (default: the library's)")
    ap.add_argument("--cheb-degree", type=int, default=0, help="override the Chebyshev preconditioner degree (experiments)")
    ap.add_argument("--precond", type=int, default=None, choices=[0, 1, 2, 3, 4, 5],
                    help="xpic_set_preconditioner kind (default: the library's)")
    ap.add_argument("--scheme", default="ecsim", choices=["ecsim", "ecsimcorr", "basic"],
                    help="ecsim is the headline workload (BASELINE configs[2]); basic = configs[1], ecsimcorr = configs[4] "
                         "at one GPU's share: side measurements")
    args = ap.parse_args(argv)
    side = args.scheme != "ecsim"
    if args.grid is None:
        args.grid = 128 if side else 256
    if args.ppc is None:
        args.ppc = 32 if side else 64
    if args.dt is None:
        args.dt = 0.1 if args.scheme == "basic" else 1.0
    if args.drift is None:
        args.drift = 0.2 if args.scheme == "basic" else 0.0
    args.n3 = tuple(args.grid_xyz) if args.grid_xyz else (args.grid,) * 3
    return args


# ---------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks.  The parent only counts devices (no HIP context on this image) and
# waits; it never initialises the GPU and never execs.  Mirrors `mpiexec -np N ... -da_processors_z N`
# (tests/ecsim/CMakeLists.txt:15-17 of the reference).
# ---------------------------------------------------------------------------------------------------------
def count_gpus():
    """Number of GPU agents of this machine WITHOUT loading the HIP runtime in this process: the KFD topology in sysfs
    (a node with simd_count > 0 is a GPU), narrowed by HIP_/ROCR_VISIBLE_DEVICES.  Falls back to a short-lived child that
    asks torch (the child may initialise HIP; the launcher itself must not: its children are fresh processes, but a parent
    holding a HIP context would sit on the card next to them)."""
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, "properties")):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
    except (OSError, ValueError):
        n = None
    if n is None:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             text=True, timeout=300)
        n = int(out.stdout.strip() or 0) if out.returncode == 0 else 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "") != "":
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip() != ""]))
    return n


def slabs_fit(args):
    n, nz = args.gpus, args.n3[2]
    if nz % n or nz // n < 6:
        print(f"bench.py --gpus {n}: the {args.n3[0]} x {args.n3[1]} x {nz} box cannot be cut into {n} z-slabs of >= 6 planes",
              file=sys.stderr, flush=True)
        return False
    return True


def launch_ranks(args, argv):
    import signal
    import tempfile

    n = args.gpus
    rehearsal = os.environ.get("XPIC_BENCH_COMM", "rccl") == "gloo"
    ndev = count_gpus()
    need = 1 if rehearsal else n
    if ndev < need:
        print(f"bench.py --gpus {n}: needs {need} MI355X device(s), found {ndev}; the xpic HIP path has no CPU "
              f"fallback and will not run fewer ranks than asked", file=sys.stderr, flush=True)
        return 3
    if not slabs_fit(args):
        return 3
    # rendezvous through a file in a private directory: no port to pick, free and hand over (bind/close/reuse is a race)
    rdv = tempfile.mkdtemp(prefix="xpic_bench_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   XPIC_BENCH_RDV=os.path.join(rdv, "store"))
        env.pop("MASTER_PORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))

    def stop_all(grace=10.0):
        """terminate our own children (exact PIDs), give them `grace` seconds, then kill what is left"""
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + grace
        for q in procs:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
        for q in procs:
            try:
                q.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    stopping = []

    def on_signal(signum, frame):
        stopping.append(signum)

    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGINT, signal.SIGTERM)}
    rc = 0
    deadline = time.monotonic() + args.rank_timeout
    try:
        while any(q.poll() is None for q in procs):
            if stopping:
                print(f"bench.py: signal {stopping[0]}: stopping the ranks", file=sys.stderr, flush=True)
                stop_all()
                rc = 128 + stopping[0]
                break
            if time.monotonic() > deadline:
                print(f"bench.py: the {n} ranks did not finish within {args.rank_timeout:.0f} s (a rank stuck in a "
                      f"collective?): stopping them", file=sys.stderr, flush=True)
                stop_all()
                rc = 124
                break
            bad = [(r, q.returncode) for r, q in enumerate(procs) if q.poll() is not None and q.returncode != 0]
            if bad:
                r, code = bad[0]
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                stop_all()
                rc = code if code > 0 else 1
                break
            time.sleep(0.2)
        else:
            bad = [q.returncode for q in procs if q.returncode != 0]
            rc = 0 if not bad else (bad[0] if bad[0] > 0 else 1)
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        try:
            for f in os.listdir(rdv):
                os.unlink(os.path.join(rdv, f))
            os.rmdir(rdv)
        except OSError:
            pass
    return rc


def cpu_baseline(args):
    """Times the CPU oracle (kind "port": the reference-faithful restatement) on a bounded sample of the same
    workload on this host's cores: same scheme, same ppc, smaller grid."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    n = args.cpu_grid
    ppc = args.ppc
    threads = oracle_lib.default_threads()
    o = oracle_lib.OracleSim(args.scheme, (n, n, n), (args.dx,) * 3, args.dt)
    rng = np.random.default_rng(1)
    npart = 0
    for i, (Np, dens, q, m) in enumerate(species(args)):
        s = o.add_sort(Np, dens, q, m)
        k = Np * n ** 3
        pts = np.empty((k, 6))
        pts[:, :3] = rng.random((k, 3)) * (n * args.dx)
        v = rng.normal(0, args.vth, (k, 3))
        v[:, 0] += beam_drift(args, i)
        pts[:, 3:] = v / np.sqrt(1.0 + (v * v).sum(1, keepdims=True))
        o.add_particles(s, pts)
        npart += k
        del pts, v
    B = np.zeros(o.fshape())
    B[..., 2] = args.b0
    o.set_field("B", B)
    o.set_field("B0", B)
    o.step()  # warm-up (first touch, allocator)
    o.solve_stats(reset=True)
    t0 = time.perf_counter()
    nsteps = args.cpu_steps
    for _ in range(nsteps):
        o.step()
    dt = time.perf_counter() - t0
    ksp_s, its = o.solve_stats()
    solver = "GMRES(30), no preconditioner" if args.scheme != "basic" else "none (explicit scheme)"
    return {
        "ms_per_solve_at_sample_grid": (ksp_s / nsteps * 1e3 / (2 if args.scheme == "ecsimcorr" else 1)) if its else None,
        "value": npart * nsteps / dt,
        "unit": "particles/s",
        "cores": threads,  # OpenMP threads actually used = the CPUs this process has (affinity mask and cgroup quota)
        "host_logical_cpus": os.cpu_count(), "affinity_cpus": oracle_lib.affinity_cpus(),
        "cgroup_cpus": oracle_lib.cgroup_cpus(),
        "threads_capped_by": "XPIC_ORACLE_THREADS" if os.environ.get("XPIC_ORACLE_THREADS") else
                             ("cgroup quota / affinity mask" if oracle_lib.cgroup_cpus() is not None or
                              oracle_lib.affinity_cpus() <= 2 * oracle_lib.POOL_CPU_SHARE else
                              f"pool CPU share of a one-GPU box ({oracle_lib.POOL_CPU_SHARE}): the affinity mask shows the whole host"),
        "kind": "port",
        "sample": f"oracle {args.scheme} step, {n}^3 cells x {ppc} ppc ({npart} particles), {nsteps} steps, "
                  f"{its} Krylov iterations ({solver}), {dt:.1f} s on {threads} OpenMP threads",
        "ksp_iters_per_s_at_sample_grid": (its / ksp_s) if its and ksp_s > 0 else None,
        "ksp_rows_per_s": (its * 3 * n ** 3 / ksp_s) if its and ksp_s > 0 else None,  # grid-size independent rate
        "ksp_iterations_per_step": its / nsteps,
    }


def beam_drift(args, k):
    """momentum (m c, along x) of species k: +-drift for the two beams of the basic scheme's two-stream set-up"""
    return (args.drift if k % 2 == 0 else -args.drift) if args.scheme == "basic" and args.ppc % 2 == 0 else 0.0


LOADER_PARAM = {"gradient": (4.0,), "blob": (0.01, 4.0)}


def species(args):
    """(Np, n, q, m) of the species the workload loads.  basic = BASELINE configs[1], the two-stream set-up: two electron
    species of ppc / 2 each (SURVEY 8d Config 2), counter-streaming with momentum +-drift along x -- the reference's config
    surface has no drift (simulation.tpp:24-41 never reads px): an extension of this build's loaders, see beam_drift."""
    if args.scheme == "basic" and args.ppc % 2 == 0:
        return [(args.ppc // 2, 0.5, -1.0, 1.0), (args.ppc // 2, 0.5, -1.0, 1.0)]
    return [(args.ppc, 1.0, -1.0, 1.0)]


PMC_FILES = {"ecsim": "pmc_traffic_256.txt", "basic": "pmc_traffic_basic_128.txt", "ecsimcorr": "pmc_traffic_ecsimcorr_128.txt"}


def pmc_traffic(scheme, n3, kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, KiB units, FETCH_SIZE x 2 on gfx950 -- tools/pmc_summary.py).  The file carries the hash of the
    kernel sources it was taken on (`# csrc-hash`): quoted only for that grid and scheme AND while xpic_amd/csrc still
    hashes to the same value; otherwise null (counters are not collected inside this run)."""
    import glob

    from xpic_amd import csrc_hash

    want = 256 if scheme == "ecsim" else 128
    if tuple(n3) != (want,) * 3:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + PMC_FILES[scheme])))
    if not files:
        return None
    stamp, val = None, None
    for line in open(files[-1]):
        if line.startswith("# csrc-hash"):
            stamp = line.split()[2]
        elif line.startswith(kernel) and val is None:
            val = float(line.split()[-2]) * 1e9
    return val if stamp == csrc_hash() else None


class Job:
    """How the ranks of one run meet outside the data path (timing barrier, MAX / SUM of a scalar) and which transport
    a context gets: "single"; "rccl" (production: one process per GPU); "gloo" (rehearsal: one process per rank, all
    on GPU 0, host-staged exchange); "threads" (rehearsal: one THREAD per rank in this process, all on GPU 0 -- the
    only way to 8 slabs on a pool box, which admits 6 processes to its card)."""

    def __init__(self, kind, rank, world, ring=None):
        self.kind, self.rank, self.world, self.ring = kind, rank, world, ring
        self.note = {"gloo": " [gloo rehearsal: all ranks share one GPU]",
                     "threads": " [rehearsal: one thread per rank, all ranks share one GPU]"}.get(kind, "")

    def attach(self, ctx):
        if self.kind == "rccl":
            from xpic_amd.parallel import init_rccl

            init_rccl(ctx)
        elif self.kind == "gloo":
            from xpic_amd.parallel import GlooRing

            GlooRing().attach(ctx)
        elif self.kind == "threads":
            self.ring.attach(ctx, self.rank)

    def barrier(self):
        if self.kind in ("rccl", "gloo"):
            import torch.distributed as dist

            dist.barrier()
        elif self.kind == "threads":
            self.ring.bar.wait()

    def reduce(self, x, op):
        """MAX or SUM of one float over the ranks"""
        if self.kind in ("rccl", "gloo"):
            import torch
            import torch.distributed as dist

            t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.kind == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
            return float(t.item())
        if self.kind == "threads":
            self.ring.red[self.rank] = float(x)
            self.ring.bar.wait()
            vals = list(self.ring.red)
            self.ring.bar.wait()
            return max(vals) if op == "max" else sum(vals)
        return float(x)


def run_threads(args):
    """XPIC_BENCH_COMM=threads: the N ranks of `--gpus N` as N threads of THIS process (ctypes releases the GIL inside
    the library), each with its own context, stream and slab on GPU 0."""
    import threading

    import torch
    from xpic_amd.parallel import ThreadRing

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the xpic HIP path has no CPU fallback")
    if not slabs_fit(args):
        raise SystemExit(3)
    torch.cuda.set_device(0)
    ring = ThreadRing(args.gpus)
    errs = []

    def body(rank):
        try:
            rank_body(args, rank, args.gpus, 0, Job("threads", rank, args.gpus, ring))
        except BaseException as e:  # noqa: BLE001 -- a rank that dies must not leave the others at a barrier
            errs.append((rank, e))
            ring.bar.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(args.gpus)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        first = [e for e in errs if not isinstance(e[1], threading.BrokenBarrierError)] or errs
        raise SystemExit(f"rank {first[0][0]} failed: {first[0][1]!r}")


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    mode = os.environ.get("XPIC_BENCH_COMM", "rccl")
    if mode not in ("rccl", "gloo", "threads"):
        raise SystemExit(f"XPIC_BENCH_COMM={mode}: rccl, gloo or threads")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if mode == "threads":
            return run_threads(args)
        sys.exit(launch_ranks(args, argv))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: refusing to report a run "
                         f"with a different number of ranks than asked")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the xpic HIP path has no CPU fallback")
    # XPIC_BENCH_COMM=gloo: rehearse the N > 1 path on a one-GPU box (all ranks share GPU 0, host-staged exchange)
    rehearsal = mode == "gloo"
    if mode == "threads" and world > 1:
        raise SystemExit("XPIC_BENCH_COMM=threads runs its ranks inside one process: start it without a launcher")
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdv = os.environ.get("XPIC_BENCH_RDV")  # set by launch_ranks: file rendezvous; under torch.distributed.run: env://
        kw = dict(init_method="file://" + rdv) if rdv else {}
        # stdout carries exactly ONE line (rank 0's JSON): the transports' own connection chatter goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world, **kw)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), **kw)
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    rank_body(args, rank, world, local_rank, Job("single" if world == 1 else mode, rank, world))
    if world > 1:
        dist.destroy_process_group()


def rank_body(args, rank, world, local_rank, job):
    import torch
    import numpy as np
    import xpic_amd as X

    n3 = args.n3
    n = n3[0]
    cubic = n3[0] == n3[1] == n3[2]
    gname = f"{n}^3" if cubic else f"{n3[0]} x {n3[1]} x {n3[2]}"
    # N > 1: the SAME global grid, cut into z-slabs (BASELINE.json configs[3]); one slab, one process, one GPU
    ctx = X.Context(args.scheme, n3, (args.dx,) * 3, args.dt, device=local_rank, rank=rank, nranks=world)
    job.attach(ctx)
    comm_ranks = ctx.comm_size()  # read back from the communicator itself (ncclCommCount)
    if comm_ranks != world:
        raise SystemExit(f"the communicator holds {comm_ranks} ranks, expected {world}")
    N = ctx.N  # local cells
    npart = args.ppc * N
    sorts = []
    for k, (Np, dens, q, m) in enumerate(species(args)):
        s = ctx.add_sort(Np, dens, q, m, capacity=int(Np * N * 1.02) + 1024)
        ctx.load_synthetic(s, Np, args.vth, seed=1234 + rank + 7919 * k, profile=args.loader,
                           drift=(beam_drift(args, k), 0.0, 0.0), param=LOADER_PARAM.get(args.loader, (0.0, 0.0)))
        sorts.append(s)
    occupancy = ctx.occupancy(sorts[0])

    # SetMagneticField(SetUniformField): B = B0 = (0, 0, b0)
    B = np.zeros(ctx.fshape())
    B[..., 2] = args.b0
    ctx.set_field(X.B, B)
    ctx.set_field(X.B0, B)
    del B

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        job.barrier()

    if args.scheme != "basic" and args.fill_kernel is not None:
        ctx.set_fill_kernel(args.fill_kernel)
    if args.fused_rebin is not None:
        ctx.set_fused_rebin(args.fused_rebin)
    if args.plain_gmres and args.scheme != "basic":
        ctx.set_preconditioner(0)
    elif (args.cheb_degree > 0 or args.precond is not None) and args.scheme != "basic":
        ctx.set_preconditioner(args.precond if args.precond is not None else 1, args.cheb_degree)
    copy_rate = ctx.probe_copy_bandwidth(1 << 30, 5) if args.probe else None
    for _ in range(args.warmup):
        ctx.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.comm_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        its += ctx.step()
    barrier()
    elapsed = time.perf_counter() - t0
    comm = ctx.comm_stats()
    elapsed = job.reduce(elapsed, "max")
    its_total = job.reduce(its, "sum")

    cg_line = None
    if args.scheme == "basic" and world == 1:
        # BASELINE.json configs[1] asks for "Poisson CG": the reference has no Poisson solve; its SPD operator is
        # matM = 2 I + 0.5 dt^2 rot- rot+ (Config 2) -> CG on matM with a manufactured right-hand side
        rng = np.random.default_rng(7)
        kctx = X.Context("ecsim", n3, (args.dx,) * 3, args.dt, device=local_rank)  # owns the Krylov workspace
        kctx.set_preconditioner(0)  # CG is unpreconditioned: no flexible-GMRES workspace
        kctx.set_field(X.W0, rng.normal(0.0, 1.0, kctx.fshape()))
        kctx.solve(X.OP_MATM_CG, X.W0, X.W1, 1e-10, 1e-50, 500)
        kctx.synchronize()
        t1 = time.perf_counter()
        cg_its = 0
        for _ in range(3):
            it, reason, rn = kctx.solve(X.OP_MATM_CG, X.W0, X.W1, 1e-10, 1e-50, 500)
            cg_its += it
        kctx.synchronize()
        dt_cg = time.perf_counter() - t1
        kctx.close()
        cg_line = {"operator": "matM (13-point, SPD), matrix-free", "iterations": cg_its / 3, "rtol": 1e-10,
                   "iters_per_s": cg_its / dt_cg, "algorithmic_GBps": 11 * 24 * N * cg_its / dt_cg / 1e9}

    prof = {k: ctx.profile_get(k) for k in ("matA_apply", "solve_matA", "fill_current", "move_bin", "scatter",
                                            "second_push", "mdot", "maxpy", "matL_zero", "scan", "index", "rot_apply",
                                            "halo", "migrate", "matL_ghost_rows", "basic_push", "corr_first_push", "precond_setup",
                                            "corr_second_push", "solve_matM", "precond", "matL_apply", "rebin",
                                            "allreduce", "cheb_steps", "rebuild_keys", "fill_gather", "precond_fallback", "precond_scaled", "precond_probation")}
    COUNTERS = ("allreduce", "cheb_steps", "rebuild_keys", "fill_gather", "precond_fallback", "precond_scaled", "precond_probation")
    count_local = sum(ctx.count(s) for s in sorts)
    count = int(job.reduce(count_local, "sum"))
    assert count == world * npart, "particles were lost in a periodic box"

    headline = {"ecsim": "particles pushed/sec (ECSIM full step) + KSP iters/sec, 256^3 grid 64ppc",
                "basic": "particles pushed/sec (basic full step: Boris + Esirkepov + FDTD)",
                "ecsimcorr": "particles pushed/sec (ecsimcorr full step) + KSP iters/sec"}[args.scheme]
    workload = {
        "ecsim": f"3D ECSIM electromagnetic, {gname} cells, {args.ppc} ppc, 1 electron species, GMRES(30) on matL+matM "
                 f"rtol=atol=1e-7 (BASELINE.json configs[2])",
        "basic": f"3D explicit (basic) scheme, {gname} cells, {len(sorts)} electron species x {args.ppc // len(sorts)} ppc "
                 f"({'two-stream set-up: beams of momentum +-%g m c along x' % args.drift if args.drift else 'both thermal'}), "
                 f"Boris push + Esirkepov deposit + FDTD, and CG on matM as the SPD solve "
                 f"(BASELINE.json configs[1]; side measurement)",
        "ecsimcorr": f"3D ecsimcorr charge-conserving scheme, {gname} cells, {args.ppc} ppc, two Esirkepov deposits + two "
                     f"solves per step (one GPU's share of BASELINE.json configs[4]; side measurement)",
    }[args.scheme]
    ms_solve = prof["solve_matA"][1] + prof["solve_matM"][1]
    line = {
        "metric": headline,
        "value": world * npart * args.steps / elapsed,  # every rank holds npart particles of the one global box
        "unit": "particles/s",
        "n_gpus": comm_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",  # the box is fixed; N GPUs cut it into N z-slabs (BASELINE.json configs[2] / [3])
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "grid": list(n3), "ppc": args.ppc, "species": len(sorts), "particles_per_gpu": npart, "dx": args.dx, "dt": args.dt,
            "loader": {"poisson": "uniform over the box (Poisson cell occupancy, as CoordinateInBox)",
                       "regular": "exactly ppc particles in every cell",
                       "gradient": "density falling linearly 4 : 1 along x (same particle total)",
                       "blob": "1 % of the particles in a Gaussian clump of sigma = 4 cells at the centre of the box, the rest "
                               "uniform (same particle total)"}[args.loader],
            "parallelism": "1 GPU" if world == 1 else
                           f"{world} z-slabs of {n3[2] // world} planes, RCCL halo / migration / dot all-reduce over xGMI"
                           + job.note,
        },
        "ksp_iters_per_s": its_total / world / (ms_solve * 1e-3) if ms_solve else None,  # iterations are global
        "ksp_method": None if args.scheme == "basic" else (
            "GMRES(30), no preconditioner" if args.plain_gmres or args.precond == 0 else
            "flexible GMRES(30), right-preconditioned by a Chebyshev polynomial in " +
            ("matM + diag(r) <matL> (the translation average of the assembled mass matrix, its rows scaled by the local "
             "density ratio r: one 123-point stencil, fp32)" if args.precond == 4 or (args.precond in (None, 5) and prof["precond_scaled"][0]) else
             "matM + <matL> (the translation average of the assembled mass matrix: one constant 123-point stencil, fp32)"
             if args.precond in (None, 3, 5) else "matM (fp32 work vectors)" if args.precond == 1 else "matM (fp64)") +
            "; outer iterations, each = 1 matA apply + the polynomial's stencil applies"),
        "ksp_iterations_per_step": its_total / world / args.steps,
        # "KSP iterations" of two methods are not one unit (an outer iteration here carries the polynomial's stencil steps, the
        # CPU baseline's are plain GMRES iterations): the comparable figures are the time per converged solve and what it holds
        "solve_ms_per_step": ms_solve / args.steps,
        "solves_per_step": (prof["solve_matA"][0] + prof["solve_matM"][0]) / args.steps,
        "ms_per_solve": ms_solve / max(1, prof["solve_matA"][0] + prof["solve_matM"][0]),
        "matA_applies_per_step": prof["matA_apply"][0] / args.steps,
        "stencil_steps_per_iteration": (prof["cheb_steps"][0] / (its_total / world)) if its_total else 0,
        "phase_ms_per_step": {k: v[1] / args.steps for k, v in prof.items() if k not in COUNTERS},
        # how the particle kernels met this load (xpic_sort_occupancy of the first species before the timed steps; the
        # fall-backs counted inside them): a cell above 64 / 128 particles costs the assembly a second / third staging pass, a
        # cell above the bucket capacity sends the step through the index pass, the largest x-pencil sets the duration of its
        # colour launch
        "occupancy": dict(occupancy, mean_pencil=npart / len(sorts) / (n3[1] * (n3[2] // world)),
                          index_passes_per_step=prof["index"][0] / args.steps,
                          key_rebuilds_per_step=prof["rebuild_keys"][0] / args.steps,
                          gathering_assemblies_per_step=prof["fill_gather"][0] / args.steps,
                          precond_fallbacks_per_step=prof["precond_fallback"][0] / args.steps,
                          density_scaled_surrogates_per_step=prof["precond_scaled"][0] / args.steps,
                          surrogates_on_probation_per_step=prof["precond_probation"][0] / args.steps),
        "allreduces_per_step": prof["allreduce"][0] / args.steps,  # reductions that are all-reduces on slabs (counted on 1 GPU too)
        "device_copy_GBps": copy_rate / 1e9 if copy_rate else None,
        # what ONE rank puts on its links per step (rank 0; every slab sends the same): point-to-point messages to the two
        # z-neighbours (halos, particle migration, the matL ghost rows) and the all-reduces of the Krylov dot products
        "comm_messages_per_step": comm[0] / args.steps if world > 1 else 0,
        "comm_bytes_per_step": comm[1] / args.steps if world > 1 else 0,
        "comm_allreduces_per_step": comm[2] / args.steps if world > 1 else 0,
        "comm_allreduce_bytes_per_step": comm[3] / args.steps if world > 1 else 0,
        "cg_matM": cg_line,
    }

    # ---- rooflines.  `roofline` describes the kernel that takes the largest share of the step.
    n_fill, ms_fill = prof["fill_current"]  # one entry per colour launch of k_ecsim_fill
    fill = None
    if n_fill and args.scheme != "basic":
        launches_per_step = n_fill / args.steps
        avg_ms = ms_fill / n_fill
        # algorithmic flop per launch: SURVEY 8(d)'s ~1200 flop per particle (576 products of the 24 x 24 block the
        # reference fills, :149-166, + their adds + the per-particle algebra) x the particles one colour launch covers
        flop_launch = FILL_FLOP_PER_PARTICLE * count_local / launches_per_step
        tf = flop_launch / (avg_ms * 1e-3) / 1e12
        # (with the re-binning's scatter deferred into it -- ctx fused_rebin, ecsim on one slab -- the launch also reads
        # the 4-byte source index and writes the 48-byte sorted record of every particle)
        # (whether it did: the library counts its gathering assemblies -- on z-slabs too since round 4; mode 2 leaves the stores
        # to the second push)
        fused = prof["fill_gather"][0] > 0 and args.fused_rebin != 2
        bytes_launch = ((100.0 if fused else 48.0) + 2952.0 / args.ppc) * count_local / launches_per_step
        gbs_fill = bytes_launch / (avg_ms * 1e-3) / 1e9
        # the binding roof by the ridge test: flop per algorithmic byte against peak flop / peak bytes (78.6 T / 8 T = 9.8)
        intensity, ridge = flop_launch / bytes_launch, FP64_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
        hbm_bound = intensity < ridge
        fill = {
            "kernel": ("k_ecsim_fill_ws" if ctx.fill_variant()[2] else "k_ecsim_fill") + " (mass matrix + currI" +
                      (" + the re-binning's gather, move and sorted copy" if fused else "") + "; one colour launch)",
            "bound": "hbm" if hbm_bound else "mfma",
            "achieved": gbs_fill if hbm_bound else tf, "peak": HBM_PEAK_GBS if hbm_bound else FP64_PEAK_TF,
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": gbs_fill / HBM_PEAK_GBS if hbm_bound else tf / FP64_PEAK_TF,
            "frac_fp64": tf / FP64_PEAK_TF, "frac_hbm": gbs_fill / HBM_PEAK_GBS,
            "flop_per_byte": intensity, "ridge_flop_per_byte": ridge, "achieved_TFLOPs": tf, "achieved_GBps": gbs_fill,
            "traffic": pmc_traffic(args.scheme, n3, "k_ecsim_fill_ws" if ctx.fill_variant()[2] else "k_ecsim_fill<") if world == 1 else None,
            "flop_per_particle": FILL_FLOP_PER_PARTICLE, "flop_per_launch": flop_launch,
            "launches": n_fill, "launches_per_step": launches_per_step, "avg_ms": avg_ms,
            "ms_per_assembly": ms_fill / args.steps,
            "bytes_per_launch": bytes_launch,
        }
    n_apply, ms_apply = prof["matA_apply"]
    spmv = None
    if n_apply:
        # algorithmic bytes of one matA apply: 123 fp64 coefficients per row, 3N rows, + read x + write y
        bytes_apply = (123 * 3 * 8 + 2 * 24) * N  # per GPU: its own slab
        achieved = bytes_apply / (ms_apply / n_apply * 1e-3) / 1e9
        spmv = {
            "kernel": "k_matA (matL+matM SpMV)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args.scheme, n3, "k_matA<true>") if world == 1 else None,
            "bytes_per_launch": bytes_apply, "launches": n_apply, "avg_ms": ms_apply / n_apply,
        }
    esk = {}
    for key, kern, bpp, bpc in (("basic_push", "k_esirkepov_push<0", 96.0, 72.0),
                                ("corr_first_push", "k_esirkepov_push<1", 72.0, 24.0),
                                ("corr_second_push", "k_esirkepov_push<2", 96.0, 72.0)):
        nl, ms = prof[key]
        if not nl:
            continue
        # R 48 + W 48 B per particle (first_push of ecsimcorr writes positions only: W 24) + the E/B
        # read and J write of a cell, once per cell
        bytes_launch = bpp * count_local / len(sorts) + bpc * N  # one launch per species
        gbs = bytes_launch / (ms / nl * 1e-3) / 1e9
        esk[key] = {
            "kernel": f"{kern}, ...> ({key})", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(args.scheme, n3, kern) if world == 1 else None,
            "bytes_per_launch": bytes_launch, "bytes_per_particle": bpp, "launches": nl, "avg_ms": ms / nl,
            "particles_per_s": count_local / len(sorts) / (ms / nl * 1e-3),
        }
    if args.scheme == "ecsim":
        line["roofline"] = fill
        line["roofline_spmv"] = spmv
        # The whole step against the HBM roofline: SURVEY 8(d)'s algorithmic bytes of one ecsim step = 334 B per particle
        # (first_push 72 + assembly 48 + second_push 72 + re-binning 96, + 2952 B of matL per cell) + the Krylov solve,
        # GMRES iteration j of a cycle = one matA apply (3000 N) + (2 j + 9) vectors of 24 N bytes
        its_step = its_total / world / args.steps
        full, frac_it = int(its_step), its_step - int(its_step)
        it_bytes = lambda j: 3000.0 * N + (2 * j + 9) * 24.0 * N
        solve_bytes = sum(it_bytes(j % 30) for j in range(full)) + frac_it * it_bytes(full % 30)
        step_bytes = (72 + 48 + 72 + 96 + 2952.0 / args.ppc) * count_local + solve_bytes
        gbs_step = step_bytes / (elapsed / args.steps) / 1e9
        line["roofline_step"] = {
            "what": "SURVEY 8(d) algorithmic HBM bytes of one whole step (per GPU) / ms_per_step", "bound": "hbm",
            "achieved": gbs_step, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_step / HBM_PEAK_GBS,
            "bytes_per_step": step_bytes, "particle_bytes": step_bytes - solve_bytes, "solve_bytes": solve_bytes,
            "frac_of_device_copy_rate": gbs_step / (copy_rate / 1e9) if copy_rate else None,
        }
        # algorithmic HBM bytes per particle and step of the ecsim particle phases: first_push 72 +
        # assembly 48 (+ 2952 B of matL per cell) + second_push 72 + re-binning 96
        ms_part = sum(prof[k][1] for k in ("fill_current", "second_push", "move_bin", "scatter", "scan", "index", "rebin")) / args.steps
        bpp = 72 + 48 + 72 + 96 + 2952.0 / args.ppc
        gbs = bpp * count_local / (ms_part * 1e-3) / 1e9
        line["roofline_particles"] = {
            "kernels": "k_ecsim_fill + k_second_push + re-binning (all particle phases of a step)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_particle": bpp, "ms_per_step": ms_part,
            "particles_per_s": count_local / (ms_part * 1e-3),
        }
    elif args.scheme == "basic":
        line["roofline"] = esk.get("basic_push")
    else:
        # ecsimcorr: the two Esirkepov passes and the assembly; `roofline` = whichever takes longer per step
        cands = [r for r in (esk.get("corr_second_push"), esk.get("corr_first_push")) if r]
        per_step = lambda r: r["avg_ms"] * r["launches"] / args.steps
        best = max(cands, key=per_step) if cands else None
        if fill and (best is None or fill["ms_per_assembly"] > per_step(best)):
            line["roofline"] = fill
            line["roofline_esirkepov"] = esk
        else:
            line["roofline"] = best
            line["roofline_esirkepov"] = esk
            line["roofline_assembly"] = fill
        line["roofline_spmv"] = spmv
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
