"""Worker of tests/test_comm_gloo.py: exercises xpic_amd.parallel.GlooRing on CPU (no GPU, no library calls)."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xpic_amd.parallel import GlooRing, neighbours, slab  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, n = dist.get_rank(), dist.get_world_size()
    ring = GlooRing()
    lo, hi = neighbours(rank, n)
    # halo-like exchange: equal sizes; payload identifies (sender, direction)
    down = np.full(1000, 10 * rank + 1, dtype=np.float64).tobytes()
    up = np.full(1000, 10 * rank + 2, dtype=np.float64).tobytes()
    fu, fd = ring.sendrecv(down, up, len(down), len(up))
    assert np.all(np.frombuffer(fu, dtype=np.float64) == 10 * hi + 1), "from_up must be the upper neighbour's down message"
    assert np.all(np.frombuffer(fd, dtype=np.float64) == 10 * lo + 2), "from_down must be the lower neighbour's up message"
    # migration-like exchange: ragged sizes, counts first (as sort_rebin does), one direction empty
    cnt_down, cnt_up = 3 + rank, 0 if rank == 0 else 5
    cfu, cfd = ring.sendrecv(np.int32(cnt_down).tobytes(), np.int32(cnt_up).tobytes(), 4, 4)
    n_fu, n_fd = int(np.frombuffer(cfu, dtype=np.int32)[0]), int(np.frombuffer(cfd, dtype=np.int32)[0])
    assert n_fu == 3 + hi and n_fd == (0 if lo == 0 else 5)
    pd = np.arange(cnt_down * 6, dtype=np.float64) + 1000 * rank
    pu = np.arange(cnt_up * 6, dtype=np.float64) - 1000 * rank
    fu, fd = ring.sendrecv(pd.tobytes(), pu.tobytes(), n_fu * 48, n_fd * 48)
    assert np.array_equal(np.frombuffer(fu, dtype=np.float64), np.arange(n_fu * 6, dtype=np.float64) + 1000 * hi)
    assert np.array_equal(np.frombuffer(fd, dtype=np.float64), np.arange(n_fd * 6, dtype=np.float64) - 1000 * lo)
    # Krylov dot products
    a = np.arange(31, dtype=np.float64) * (rank + 1)
    ring.allreduce_sum(a)
    assert np.array_equal(a, np.arange(31, dtype=np.float64) * (n * (n + 1) // 2))
    # slab bookkeeping covers the grid exactly once
    z0, nzl = slab(24, rank, n)
    cover = np.zeros(24)
    cover[z0:z0 + nzl] = 1
    ring.allreduce_sum(cover)
    assert np.all(cover == 1)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{n} ok", flush=True)


if __name__ == "__main__":
    main()
