"""bench.py's N > 1 entry: `--gpus N` must start N ranks (or fail loudly), never quietly run one.
The reference's analogue is `mpiexec -np 2 ... -da_processors_x 2` (tests/ecsim/CMakeLists.txt:15-17)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **kw)
    return env


def test_gpus_2_without_devices_fails_cleanly():
    """No GPU in this container: the launcher must refuse (non-zero, message, no JSON line), not fall back to one rank."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "32", "--ppc", "8", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=_env(XPIC_BENCH_COMM="gloo"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs 1 MI355X device" in out.stderr and "will not run fewer ranks" in out.stderr
    assert "n_gpus" not in out.stdout


def test_world_size_mismatch_is_refused():
    """Started as ONE rank by a launcher but asked for --gpus 2: refuse instead of reporting n_gpus = 1."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "32", "--ppc", "8", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
    assert "n_gpus" not in out.stdout


def test_launcher_does_not_touch_the_gpu_or_exec():
    """The parent of the N ranks only counts devices: no torch.cuda.is_available / set_device / os.exec* in launch_ranks."""
    import inspect

    sys.path.insert(0, ROOT)
    import bench

    src = inspect.getsource(bench.launch_ranks) + inspect.getsource(bench.count_gpus)
    assert "is_available" not in src and "set_device" not in src and "os.exec" not in src and "Context(" not in src
    assert "import torch\n" not in src  # the HIP runtime is never loaded by the launcher itself
    assert "Popen" in src and "kill()" in src and "rank_timeout" in src
    # the launcher process must really stay off the GPU runtime: counting devices loads neither torch nor libamdhip64
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.count_gpus(); "
            "maps = open('/proc/self/maps').read(); "
            "assert 'libamdhip64' not in maps and 'libtorch' not in maps, 'GPU runtime loaded'; print('gpus', n)" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("gpus"), out.stdout + out.stderr


def test_launcher_stops_ranks_that_hang():
    """A rank stuck in a collective must not hang the parent: on the wall-clock limit the launcher terminates (then kills)
    the children it started and exits non-zero.  Here the 'ranks' are stand-ins that ignore SIGTERM and sleep."""
    code = r"""
import os, sys, time
sys.path.insert(0, %r)
import bench
bench.count_gpus = lambda: 2
hang = os.path.join(%r, 'tests', '_hang_rank.py')
open(hang, 'w').write('import signal, time\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\ntime.sleep(600)\n')
real_popen = bench.subprocess.Popen
bench.subprocess.Popen = lambda cmd, env=None: real_popen([sys.executable, hang], env=env)
args = bench.parse_args(['--gpus', '2', '--grid', '32', '--rank-timeout', '2'])
t0 = time.time()
rc = bench.launch_ranks(args, [])
os.unlink(hang)
print('rc', rc, 'after', round(time.time() - t0, 1))
""" % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rc 124" in out.stdout and "did not finish within" in out.stderr


@pytest.mark.gpu
def test_two_rank_rehearsal_reports_two_ranks():
    """`bench.py --gpus 2` on the one-GPU box: two child ranks share GPU 0, exchange over gloo (XPIC_BENCH_COMM=gloo),
    the line says n_gpus = 2 (read back from the communicator) and no particle is lost."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "32", "--ppc", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=_env(XPIC_BENCH_COMM="gloo"), capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["config"]["particles_per_gpu"] == 8 * 32 * 32 * 16
    assert line["value"] > 0 and line["ksp_iterations_per_step"] > 0


@pytest.mark.gpu
def test_eight_rank_rehearsal_in_one_process():
    """BASELINE configs[3]'s rank count through bench.py itself: `--gpus 8 --grid 128` with XPIC_BENCH_COMM=threads runs
    the eight slabs (16 planes each) as eight threads of one process on GPU 0 (a pool box admits 6 processes to its card);
    the line reports the 8 ranks of the communicator, every particle is still there, the solve converges in the same
    number of iterations on every rank, and the all-reduces stay at one per Krylov iteration + the step's fixed ones."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--grid", "128", "--ppc", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=_env(XPIC_BENCH_COMM="threads"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["config"]["particles_per_gpu"] == 8 * 128 * 128 * 16
    assert "8 z-slabs of 16 planes" in line["config"]["parallelism"]
    its = line["ksp_iterations_per_step"]
    assert its > 0 and its == int(its)  # the same count on all 8 ranks (the sum over ranks / 8 is whole)
    assert line["phase_ms_per_step"]["halo"] > 0 and line["phase_ms_per_step"]["migrate"] > 0
    # one per Krylov iteration + the fixed ones of a step (norm of the right-hand side, initial residual, the surrogate's
    # sums, the migration's flags) + the explicit norms of the last iterations (residual below 1e-6 of the right-hand side:
    # krylov.hip's rule) + the assembly's error word + -- at 8 particles per cell the spread of matL's diagonal is above the
    # default preconditioner's threshold -- the largest density ratio of the scaled surrogate; classical Gram-Schmidt with
    # a separate norm would be 2 per iteration + the fixed ones
    assert line["allreduces_per_step"] <= its + 8


@pytest.mark.gpu
def test_single_rank_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, BENCH, "--grid", "32", "--ppc", "16", "--steps", "2", "--warmup", "1",
                          "--cpu-grid", "16", "--cpu-steps", "1"], env=_env(), capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["kernel"].startswith("k_ecsim_fill") and r["flop_per_particle"] == 1200.0
    # the binding roof by the ridge test: flop per algorithmic byte against 78.6 TFLOP/s / 8 TB/s; both fractions are carried
    assert r["bound"] == ("hbm" if r["flop_per_byte"] < r["ridge_flop_per_byte"] else "mfma")
    assert r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["frac"] - (r["frac_hbm"] if r["bound"] == "hbm" else r["frac_fp64"])) < 1e-12
    for key in ("solve_ms_per_step", "ms_per_solve", "matA_applies_per_step", "stencil_steps_per_iteration", "occupancy"):
        assert key in line, key
    assert line["occupancy"]["gathering_assemblies_per_step"] == 1  # the default step defers its scatter into the assembly
    assert line["roofline_spmv"]["bound"] == "hbm"
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["ksp_iters_per_s_at_sample_grid"] > 0


def test_pmc_traffic_is_quoted_only_for_the_sources_it_was_taken_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from the newest committed PMC summary, and only while xpic_amd/csrc still hashes to
    the value stamped in that file; another grid, an unknown kernel or a stale stamp give null (never a stale number)."""
    sys.path.insert(0, ROOT)
    import bench
    import xpic_amd

    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = "# csrc-hash %s   (stamp)\nkernel calls read write total ms\nk_ecsim_fill<true, true>  32  9.2  7.2  16.4  6.2\n"
    (prof / "r03_pmc_traffic_256.txt").write_text(good % xpic_amd.csrc_hash())
    assert bench.pmc_traffic("ecsim", (256, 256, 256), "k_ecsim_fill") == pytest.approx(16.4e9)
    assert bench.pmc_traffic("ecsim", (128, 128, 128), "k_ecsim_fill") is None      # not the grid it was taken on
    assert bench.pmc_traffic("ecsim", (256, 256, 256), "k_no_such_kernel") is None
    assert bench.pmc_traffic("basic", (128, 128, 128), "k_esirkepov_push<0") is None  # no file for that scheme
    (prof / "r04_pmc_traffic_256.txt").write_text(good % ("0" * 40))                 # a newer file, other sources
    assert bench.pmc_traffic("ecsim", (256, 256, 256), "k_ecsim_fill") is None


def test_workload_helpers():
    """The argument surface of bench.py that the driver and the side configurations rely on: default = BASELINE configs[2]
    on one GPU, `--scheme basic` = two species of ppc / 2, `--grid-xyz` a non-cubic box, and the slab rule of --gpus N."""
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args([])
    assert a.gpus == 1 and a.n3 == (256, 256, 256) and a.ppc == 64 and a.scheme == "ecsim"
    assert bench.species(a) == [(64, 1.0, -1.0, 1.0)]
    b = bench.parse_args(["--scheme", "basic", "--grid", "128", "--ppc", "32"])
    assert bench.species(b) == [(16, 0.5, -1.0, 1.0)] * 2
    c = bench.parse_args(["--scheme", "ecsimcorr", "--grid-xyz", "512", "512", "64", "--ppc", "32", "--gpus", "8"])
    assert c.n3 == (512, 512, 64) and bench.slabs_fit(c)          # 8 slabs of 8 planes
    d = bench.parse_args(["--grid", "32", "--gpus", "8"])
    assert not bench.slabs_fit(d)                                   # 4 planes per slab: below the 6 a slab needs
    e = bench.parse_args(["--grid", "100", "--gpus", "8"])
    assert not bench.slabs_fit(e)                                   # not divisible
