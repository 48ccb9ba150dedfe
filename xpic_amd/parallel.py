"""One process per GPU: z-slab bookkeeping and the two ways a context gets its communicator.

* `init_rccl(ctx)`  -- production: RCCL over xGMI.  torch.distributed is only the bootstrap channel that carries
  rank 0's 128-byte ncclUniqueId to the other ranks; all data-path traffic is issued by libxpic_hip.so itself.
* `ThreadRing`      -- tests / rehearsals: the same callbacks between THREADS of one process (one context per thread, all
  on one GPU): N slabs under the GPU pool's limit of 6 processes per card.
* `GlooRing`        -- tests: a host-staged ring send/receive + all-reduce over torch.distributed (gloo), plugged in
  through xpic_comm_init_callbacks.  Lets two ranks that share one GPU (or none, for the transport's own tests)
  exercise the slab code path.
"""
import threading

import numpy as np


def slab(nz, rank, nranks):
    """(first plane, number of planes) of z-slab `rank`: DMDA with da_processors_z = nranks, equal slabs."""
    if nz % nranks:
        raise ValueError("nz must be divisible by the number of z-slabs")
    nzl = nz // nranks
    return rank * nzl, nzl


def neighbours(rank, nranks):
    """(lower, upper) z-neighbour in the periodic ring."""
    return (rank - 1 + nranks) % nranks, (rank + 1) % nranks


def init_rccl(ctx, device=None):
    import torch
    import torch.distributed as dist
    import xpic_amd

    rank = dist.get_rank()
    if dist.get_backend() == "nccl":
        t = torch.zeros(128, dtype=torch.uint8, device=device or torch.device("cuda", torch.cuda.current_device()))
    else:
        t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(xpic_amd.rccl_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, src=0)
    ctx.comm_init_rccl(bytes(t.cpu().numpy().tobytes()))


def map_peers(ctx):
    """Copy-engine path of the large messages (xpic_comm_peer_export / _import): every rank publishes the blob of its
    receive buffers over torch.distributed -- the bootstrap channel, like the ncclUniqueId -- and maps its two z-neighbours'."""
    import torch.distributed as dist

    n, rank = dist.get_world_size(), dist.get_rank()
    blobs = [None] * n
    dist.all_gather_object(blobs, ctx.comm_peer_export())
    lo, hi = neighbours(rank, n)
    ctx.comm_peer_import(blobs[lo], blobs[hi])


class GlooRing:
    """Ring exchange with the semantics xpic_comm_callbacks asks for, over torch.distributed point-to-point ops.
    Messages to the same peer (2 ranks: both neighbours are one process) are told apart by tag: 0 = sent downwards,
    1 = sent upwards."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist
        self.rank = dist.get_rank()
        self.n = dist.get_world_size()
        self.lo, self.hi = neighbours(self.rank, self.n)

    def sendrecv(self, down, up, n_from_up, n_from_down):
        import torch

        dist = self.dist
        reqs = []
        keep = []
        if len(down):
            t = torch.frombuffer(bytearray(down), dtype=torch.uint8)
            keep.append(t)
            reqs.append(dist.isend(t, self.lo, tag=0))
        if len(up):
            t = torch.frombuffer(bytearray(up), dtype=torch.uint8)
            keep.append(t)
            reqs.append(dist.isend(t, self.hi, tag=1))
        fu = torch.empty(n_from_up, dtype=torch.uint8)
        fd = torch.empty(n_from_down, dtype=torch.uint8)
        if n_from_up:
            reqs.append(dist.irecv(fu, self.hi, tag=0))  # the upper neighbour's "down" message
        if n_from_down:
            reqs.append(dist.irecv(fd, self.lo, tag=1))  # the lower neighbour's "up" message
        for r in reqs:
            r.wait()
        return fu.numpy().tobytes(), fd.numpy().tobytes()

    def allreduce_sum(self, arr):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr))
        self.dist.all_reduce(t)
        arr[:] = t.numpy()

    def attach(self, ctx):
        ctx.comm_init_callbacks(self.sendrecv, self.allreduce_sum)


class ThreadRing:
    """The xpic_comm_callbacks transport between THREADS of one process: every rank's context lives in its own thread of
    this process and the ring exchange / all-reduce are rendezvous on barriers.  One process on the GPU whatever the
    number of ranks -- the way to run the 8 slabs of BASELINE configs[3] / [4] under the pool's limit of 6 processes per
    card."""

    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n, timeout=240)
        self.down = [b""] * n
        self.up = [b""] * n
        self.red = [None] * n
        self.blobs = [None] * n

    def map_peers(self, ctx, rank):
        """xpic_comm_peer_export / _import between the threads of this process (the buffers' own addresses are used)"""
        self.blobs[rank] = ctx.comm_peer_export()
        self.bar.wait()
        ctx.comm_peer_import(self.blobs[(rank - 1 + self.n) % self.n], self.blobs[(rank + 1) % self.n])
        self.bar.wait()

    def attach(self, ctx, rank):
        n = self.n
        lo, hi = (rank - 1 + n) % n, (rank + 1) % n

        def sendrecv(down, up, n_from_up, n_from_down):
            self.down[rank], self.up[rank] = bytes(down), bytes(up)
            self.bar.wait()
            fu, fd = self.down[hi], self.up[lo]  # the upper neighbour's "down" message, the lower neighbour's "up" message
            assert len(fu) == n_from_up and len(fd) == n_from_down, (rank, len(fu), n_from_up, len(fd), n_from_down)
            self.bar.wait()
            return fu, fd

        def allreduce_sum(arr):
            self.red[rank] = np.array(arr, copy=True)
            self.bar.wait()
            tot = self.red[0].copy()
            for r in range(1, n):  # the same order on every rank: bitwise the same sum everywhere
                tot += self.red[r]
            self.bar.wait()
            arr[:] = tot

        ctx.comm_init_callbacks(sendrecv, allreduce_sum)
