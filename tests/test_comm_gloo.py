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


def test_thread_ring_semantics():
    """ThreadRing (the callback transport between threads of one process, how 8 slabs run on a one-GPU box) without a GPU:
    every rank receives its upper neighbour's `down` message and its lower neighbour's `up` message, and the all-reduce
    gives every rank the same bits (a fixed summation order)."""
    import threading

    import numpy as np
    from xpic_amd.parallel import ThreadRing

    n = 8
    ring = ThreadRing(n)
    cbs = [None] * n

    class Stub:
        def __init__(self, r):
            self.r = r

        def comm_init_callbacks(self, sendrecv, allreduce_sum):
            cbs[self.r] = (sendrecv, allreduce_sum)

    for r in range(n):
        ring.attach(Stub(r), r)
    out, errs = [None] * n, []

    def body(r):
        try:
            sendrecv, allreduce = cbs[r]
            got = []
            for it in range(3):  # variable sizes, as the migration has them
                down = bytes([r, it, 0]) * (r + 1)
                up = bytes([r, it, 1]) * (2 * r + 1)
                hi, lo = (r + 1) % n, (r - 1 + n) % n
                fu, fd = sendrecv(down, up, 3 * (hi + 1), 3 * (2 * lo + 1))
                assert fu == bytes([hi, it, 0]) * (hi + 1) and fd == bytes([lo, it, 1]) * (2 * lo + 1)
                a = np.array([0.1 * (r + 1), 1e-17 * r, float(it)])
                allreduce(a)
                got.append(a.copy())
            out[r] = got
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e))
            ring.bar.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for r in range(1, n):
        for it in range(3):
            assert np.array_equal(out[r][it], out[0][it])  # bitwise the same on every rank
    assert out[0][2][2] == 2.0 * n and abs(out[0][0][0] - 0.1 * n * (n + 1) / 2) < 1e-12
