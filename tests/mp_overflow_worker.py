"""Worker of tests/test_gpu_slabs.py::test_migration_overflow_fails_on_every_rank: two z-slabs share the GPU; rank 0
holds more upward leavers than its migration buffer takes.  The error must surface on BOTH ranks (exit code 7 each),
not leave rank 1 waiting in the next exchange."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xpic_amd as X  # noqa: E402
from xpic_amd.parallel import GlooRing  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, nranks = dist.get_rank(), dist.get_world_size()
    n, d, dt = (16, 16, 8 * nranks), (0.5, 0.5, 0.25), 1.0
    ctx = X.Context("ecsim", n, d, dt, device=0, rank=rank, nranks=nranks)
    GlooRing().attach(ctx)
    s = ctx.add_sort(8, 1.0, -1.0, 1.0, capacity=200000)  # migration buffer: max(capacity / 8, 65536) records
    rng = np.random.default_rng(5 + rank)
    npart = 70000 if rank == 0 else 1000
    pts = np.empty((npart, 6))
    pts[:, 0] = rng.random(npart) * n[0] * d[0]
    pts[:, 1] = rng.random(npart) * n[1] * d[1]
    top = (ctx.z0 + ctx.nzl) * d[2]
    pts[:, 2] = top - 0.2 + 0.19 * rng.random(npart)  # the uppermost plane of the slab
    pts[:, 3:5] = 0.0
    pts[:, 5] = 0.22 if rank == 0 else 0.0             # rank 0: every particle crosses into the upper neighbour
    assert ctx.add_particles(s, pts) == npart
    ctx.ecsim_first_push(s)
    try:
        ctx.update_cells(s)  # collective
    except X.XpicError as e:
        print(f"rank {rank} error: {e}", flush=True)
        ctx.close()
        sys.exit(7)
    print(f"rank {rank} unexpectedly succeeded", flush=True)
    sys.exit(0)


if __name__ == "__main__":
    main()
