"""CPU tests of the N > 1 path's transport and bookkeeping: the ring exchange and all-reduce that the z-slab
decomposition plugs into xpic_comm_init_callbacks, run as real multi-process jobs over gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# 8 = the rank count of BASELINE configs[3] / [4]: the ring's neighbour arithmetic, tags and the all-reduce at the real size
# (the GPU side of an 8-slab run is tests/test_gpu_slabs.py::test_eight_slabs_in_one_process: 8 contexts, 8 threads, one card)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_ring_semantics(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(29611 + world), os.path.join(ROOT, "tests", "mp_ring_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" ok") == world


def test_slab_partition():
    from xpic_amd.parallel import neighbours, slab

    assert slab(256, 3, 8) == (96, 32)
    assert neighbours(0, 8) == (7, 1) and neighbours(7, 8) == (6, 0) and neighbours(0, 2) == (1, 1)
    with pytest.raises(ValueError):
        slab(10, 0, 4)
