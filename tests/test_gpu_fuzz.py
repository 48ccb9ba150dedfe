"""Random small configurations of every scheme, two steps on the GPU against the CPU oracle (tools/fuzz_steps.py): grid
extents from 6 to 40, power-of-two and other spacings, 0.3 to 70 particles per cell with cells of 64, 65, 240, 241, 300
and 700 particles and empty stretches in between, every preconditioner kind."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_configurations_match_the_oracle(oracle):
    import fuzz_steps

    worst = fuzz_steps.run(cases=18, seed=11, verbose=False)
    assert set(worst) == {"basic", "ecsim", "ecsimcorr"}
    assert worst["basic"] < 1e-12 and worst["ecsim"] < 1e-9 and worst["ecsimcorr"] < 1e-9, worst


def test_random_configurations_on_a_slab_match_the_oracle(oracle):
    """The same on a single z-slab that keeps its ghost planes and is its own neighbour over RCCL (geometry.self_ring):
    ghost exchanges, migration, the slab's colour schedule and the cleared boundary planes of the first-touch assembly,
    with empty stretches and heavy cells."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import fuzz_steps; "
            "w = fuzz_steps.run(cases=12, seed=23, verbose=False, slab=True); "
            "assert set(w) == {'basic', 'ecsim', 'ecsimcorr'}, w; "
            "assert w['basic'] < 1e-12 and w['ecsim'] < 1e-9 and w['ecsimcorr'] < 1e-9, w; print('slab fuzz ok')"
            % (os.path.join(root, "tools"), os.path.join(root, "tests")))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "slab fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
