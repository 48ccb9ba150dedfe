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
