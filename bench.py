#!/usr/bin/env python3
"""bench.py -- xpic hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" is one full ECSIM timestep (first_push + re-bin, current/mass-matrix assembly, implicit field solve,
second_push, field update) of BASELINE.json's headline configuration: 256^3 cells, 64 particles per cell,
one electron species, uniform B0 -- all resident in HBM before the timed region.  One process per GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate


def cpu_baseline(args):
    """Times the CPU oracle (kind "port": the reference-faithful restatement) on a bounded sample of the same
    workload on this host's cores: same scheme, same ppc, smaller grid."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    n = args.cpu_grid
    ppc = args.ppc
    threads = oracle_lib.default_threads()
    o = oracle_lib.OracleSim("ecsim", (n, n, n), (args.dx,) * 3, args.dt)
    s = o.add_sort(ppc, 1.0, -1.0, 1.0)
    rng = np.random.default_rng(1)
    npart = ppc * n ** 3
    pts = np.empty((npart, 6))
    pts[:, :3] = rng.random((npart, 3)) * (n * args.dx)
    v = rng.normal(0, args.vth, (npart, 3))
    pts[:, 3:] = v / np.sqrt(1.0 + (v * v).sum(1, keepdims=True))
    o.add_particles(s, pts)
    del pts, v
    B = np.zeros(o.fshape())
    B[..., 2] = args.b0
    o.set_field("B", B)
    o.set_field("B0", B)
    o.step()  # warm-up (first touch, allocator)
    t0 = time.perf_counter()
    its = 0
    nsteps = args.cpu_steps
    for _ in range(nsteps):
        its += o.step()
    dt = time.perf_counter() - t0
    return {
        "value": npart * nsteps / dt,
        "unit": "particles/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle ecsim step, {n}^3 cells x {ppc} ppc ({npart} particles), {nsteps} steps, "
                  f"{its} GMRES(30) iterations, {dt:.1f} s on {threads} OpenMP threads",
        "ksp_iters_per_s_at_sample_grid": None,
    }


def pmc_traffic(grid):
    """HBM bytes per k_matA launch from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, KiB units, FETCH_SIZE x 2 on gfx950 -- tools/pmc_summary.py); only valid for the grid it was taken on."""
    path = os.path.join(ROOT, "profiles", "r01_v8_pmc_traffic_256.txt")
    if grid != 256 or not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("k_matA<true, true>"):
            return float(line.split()[-2]) * 1e9
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--ppc", type=int, default=64)
    ap.add_argument("--dx", type=float, default=0.5)
    ap.add_argument("--dt", type=float, default=1.0)
    ap.add_argument("--vth", type=float, default=0.014)  # T = 0.1 keV electrons (tests/ecsim/ecsim_ex1.cpp:66-70)
    ap.add_argument("--b0", type=float, default=0.2)
    ap.add_argument("--cpu-grid", type=int, default=48)   # 48^3 x 64 ppc x 4 steps: ~13 s of oracle work
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe", action="store_true", help="also run the device copy probes (PMC calibration)")
    ap.add_argument("--plain-gmres", action="store_true", help="unpreconditioned GMRES(30), as the CPU oracle runs")
    ap.add_argument("--cheb-degree", type=int, default=0, help="override the Chebyshev preconditioner degree (experiments)")
    ap.add_argument("--scheme", default="ecsim", choices=["ecsim", "ecsimcorr", "basic"],
                    help="ecsim is the headline workload; the others are side measurements (no cpu_baseline)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the xpic HIP path has no CPU fallback")
    # XPIC_BENCH_COMM=gloo: rehearse the N > 1 path on a one-GPU box (all ranks share GPU 0, host-staged exchange)
    rehearsal = os.environ.get("XPIC_BENCH_COMM", "rccl") == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import xpic_amd as X

    n = args.grid
    # N > 1: the SAME global grid, cut into z-slabs (BASELINE.json configs[3]); one slab, one process, one GPU
    ctx = X.Context(args.scheme, (n, n, n), (args.dx,) * 3, args.dt, device=local_rank, rank=rank, nranks=world)
    if world > 1:
        from xpic_amd.parallel import GlooRing, init_rccl

        if rehearsal:
            GlooRing().attach(ctx)
        else:
            init_rccl(ctx)
    N = ctx.N  # local cells
    npart = args.ppc * N
    s = ctx.add_sort(args.ppc, 1.0, -1.0, 1.0, capacity=int(npart * 1.02) + 1024)
    ctx.fill_synthetic(s, args.ppc, args.vth, seed=1234 + rank)
    import numpy as np

    # SetMagneticField(SetUniformField): B = B0 = (0, 0, b0)
    B = np.zeros(ctx.fshape())
    B[..., 2] = args.b0
    ctx.set_field(X.B, B)
    ctx.set_field(X.B0, B)
    del B

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.plain_gmres and args.scheme != "basic":
        ctx.set_preconditioner(0)
    elif args.cheb_degree > 0 and args.scheme != "basic":
        ctx.set_preconditioner(1, args.cheb_degree)
    copy_rate = ctx.probe_copy_bandwidth(1 << 30, 5) if args.probe else None
    for _ in range(args.warmup):
        ctx.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        its += ctx.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dev = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        it_t = torch.tensor([its], dtype=torch.float64, device=dev)
        dist.all_reduce(it_t, op=dist.ReduceOp.SUM)
        its_total = float(it_t.item())
    else:
        its_total = float(its)

    cg_line = None
    if args.scheme == "basic" and world == 1:
        # BASELINE.json configs[1] asks for "Poisson CG": the reference has no Poisson solve; its SPD operator is
        # matM = 2 I + 0.5 dt^2 rot- rot+ (Config 2) -> CG on matM with a manufactured right-hand side
        rng = np.random.default_rng(7)
        kctx = X.Context("ecsim", (n, n, n), (args.dx,) * 3, args.dt, device=local_rank)  # owns the Krylov workspace
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
                                            "second_push", "mdot", "maxpy", "matL_zero", "scan", "rot_apply",
                                            "halo", "migrate", "matL_ghost_rows", "basic_push", "corr_first_push",
                                            "corr_second_push", "solve_matM", "precond", "matL_apply")}
    n_apply, ms_apply = prof["matA_apply"]
    n_solve, ms_solve = prof["solve_matA"]
    # algorithmic bytes of one matA apply: 123 fp64 coefficients per row, 3N rows, + read x + write y
    bytes_apply = (123 * 3 * 8 + 2 * 24) * N  # per GPU: its own slab
    achieved = bytes_apply / (ms_apply / max(n_apply, 1) * 1e-3) / 1e9 if n_apply else 0.0
    count = ctx.count(s)
    if world > 1:
        ct = torch.tensor([count], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(ct)
        count = int(ct.item())
    assert count == world * npart, "particles were lost in a periodic box"

    line = {
        "metric": "particles pushed/sec (ECSIM full step) + KSP iters/sec, 256^3 grid 64ppc",
        "value": world * npart * args.steps / elapsed,  # every rank holds npart particles of the one global box
        "unit": "particles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",  # the 256^3 box is fixed; N GPUs cut it into N z-slabs (BASELINE.json configs[2] / [3])
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"3D ECSIM electromagnetic, {n}^3 cells, {args.ppc} ppc, 1 electron species, "
                         f"GMRES(30) on matL+matM rtol=atol=1e-7 (BASELINE.json configs[2])") if args.scheme == "ecsim"
                        else f"scheme {args.scheme}, {n}^3 cells, {args.ppc} ppc (side measurement)",
            "grid": [n, n, n], "ppc": args.ppc, "particles_per_gpu": npart, "dx": args.dx, "dt": args.dt,
            "parallelism": "1 GPU" if world == 1 else
                           f"{world} z-slabs of {n // world} planes, RCCL halo / migration / dot all-reduce over xGMI",
        },
        "ksp_iters_per_s": its_total / world / (ms_solve * 1e-3) if ms_solve else None,  # iterations are global
        "ksp_method": "GMRES(30), right-preconditioned by a Chebyshev polynomial in matM" if not args.plain_gmres
                      else "GMRES(30), no preconditioner",
        "ksp_iterations_per_step": its_total / world / args.steps,
        "phase_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
        "device_copy_GBps": copy_rate / 1e9 if copy_rate else None,
        "cg_matM": cg_line,
        "roofline": {
            "kernel": "k_matA (matL+matM SpMV)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(n) if world == 1 else None,
            "bytes_per_launch": bytes_apply, "launches": n_apply, "avg_ms": ms_apply / max(n_apply, 1),
        },
    }
    n_fill, ms_fill = prof["fill_current"]
    if n_fill and args.scheme != "basic":
        # the assembly is the one dense contraction of the path: per particle a 36 x 36 rank-1 update = 2 * 1296 flop
        # (issued on the matrix cores as 9 v_mfma_f64_16x16x4_f64 per 4 particles = 4608 flop per particle with the
        # padding); fp64 matrix peak = fp64 vector peak = 78.6 TFLOP/s on MI355X (measured here: 75.5 with MFMA)
        tf = 2.0 * 1296 * count / world / (ms_fill / n_fill * 1e-3) / 1e12
        line["roofline_assembly"] = {
            "kernel": "k_ecsim_fill (mass matrix + currI, all colour launches of one assembly)", "bound": "mfma",
            "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6, "traffic": None,
            "flop_per_particle": 2592, "issued_flop_per_particle": 4608, "avg_ms": ms_fill / n_fill,
        }
    if args.scheme == "ecsim":
        # algorithmic HBM bytes per particle and step of the ecsim particle phases: first_push 72 +
        # assembly 48 (+ 2952 B of matL per cell) + second_push 72 + re-binning 96
        ms_part = sum(prof[k][1] for k in ("fill_current", "second_push", "move_bin", "scatter", "scan")) / args.steps
        bpp = 72 + 48 + 72 + 96 + 2952.0 / args.ppc
        gbs = bpp * count / world / (ms_part * 1e-3) / 1e9
        line["roofline_particles"] = {
            "kernels": "k_ecsim_fill + k_second_push + k_move_bin + k_scatter (all particle phases of a step)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_particle": bpp, "ms_per_step": ms_part,
            "particles_per_s": count / world / (ms_part * 1e-3),
        }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.scheme == "ecsim":
            line["cpu_baseline"] = cpu_baseline(args)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
