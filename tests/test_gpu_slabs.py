"""z-slab decomposition on the GPU: 2 (and 3) processes share the one GPU of the box, exchange through gloo
(xpic_comm_init_callbacks), and must reproduce the single-slab run of the same seeded problem: particle totals
exactly, fields to 1e-8 (the Krylov dot products are summed in a different order)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scheme,world", [("ecsim", 2), ("ecsim", 3), ("basic", 2), ("ecsimcorr", 2)])
def test_slabs_reproduce_single_slab(scheme, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(29700 + world), os.path.join(ROOT, "tests", "mp_slab_worker.py"), scheme]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert out.stdout.count(" ok") == world
