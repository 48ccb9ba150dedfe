"""The oracle entry points that only the `-m gpu` tests call, run on the CPU under AddressSanitizer + UBSan.

Round 3's GPU log `call_tests2` holds a segmentation fault in which >= 11 threads of the test process faulted at the same
moment.  The product library has no host threads of its own; the only team of that size in the process is the oracle's
OpenMP team, inside one of its parallel regions.  The golden-vector tests were already clean under the sanitizers; this
test puts the remaining entry points (solves, applies, phase functions, whole steps, the eccapfim kernels) under them,
at team sizes 1, 3 and 16, including the misuse sequences (apply / solve before matL was ever assembled) that index an
empty coefficient array from every thread at once.  Sanitizers run on the CPU build only (they are not available for
the GPU on this pool)."""
import ctypes as C
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SAN_DIR = os.path.join(ROOT, "oracle", "_san")
SAN_SO = os.path.join(SAN_DIR, "liboracle_san.so")
SRC = os.path.join(ROOT, "oracle", "xpic_oracle.cpp")


def _runtime(name):
    p = subprocess.run(["g++", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_gpu_only_oracle_entry_points_are_clean_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    os.makedirs(SAN_DIR, exist_ok=True)
    if not os.path.exists(SAN_SO) or os.path.getmtime(SAN_SO) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-fopenmp", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-shared", "-o", SAN_SO, SRC])
    env = dict(os.environ)
    env.update(XPIC_ORACLE_SO=SAN_SO, LD_PRELOAD=asan, PYTHONPATH=HERE,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_WAIT_POLICY="passive")
    env.pop("XPIC_ORACLE_THREADS", None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "oracle_san_driver.py")], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "SAN-DRIVER-OK" in out.stdout, (out.stdout[-2000:], out.stderr[-6000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-6000:]


def test_every_exported_oracle_function_has_its_argument_types_declared(oracle):
    """A ctypes call without argtypes passes a 64-bit handle as a C int: the truncated pointer faults inside the first
    parallel region that touches it.  Every `orc_*` symbol the library exports must be declared in oracle_lib.lib()."""
    lib = oracle.lib()
    syms = subprocess.run(["nm", "-D", "--defined-only", lib._name], capture_output=True, text=True).stdout.split("\n")
    names = sorted({ln.split()[-1] for ln in syms if ln.strip() and ln.split()[-1].startswith("orc_")})
    assert len(names) >= 50
    missing = [n for n in names if getattr(lib, n).argtypes is None and n not in ("orc_reset_rng",)]
    assert not missing, missing
    # pointer-returning / wide results must not be truncated either
    assert lib.orc_create.restype is C.c_void_p
    for n in ("orc_count", "orc_add_particles", "orc_get_particles", "orc_load_maxwell_box"):
        assert getattr(lib, n).restype is C.c_long, n
