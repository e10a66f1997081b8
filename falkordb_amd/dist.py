"""Multi-GPU BFS driver: column-slab partition + one frontier all-gather per level.

SURVEY.md §8e / BASELINE.json north_star: rank r owns destination vertices
[r*slab, (r+1)*slab) and holds A[:, slab] (push) and A'[slab, :] (pull); every level each
rank computes the new-frontier bits of its own slab (libfgpu step kernel), the slabs are
all-gathered over RCCL/xGMI (`all_gather_into_tensor`: equal 1-D slabs, so the gathered buffer
IS the global frontier bitmap), and the commit kernel adopts it.  Direction choice is
rank-local (either direction yields the same owned bits); termination is decided from the
global frontier population, identical on every rank, so no extra collective is needed.

The level loop is generic over a `backend` (begin / step / commit / done) and a `gather`
callable so that tests can drive the exact same control flow on CPU with gloo
(tests/test_dist_cpu.py supplies an oracle-backed backend; the product backend below is the
HIP one and has no CPU fallback).
"""
from __future__ import annotations

from typing import Callable


def slab_range(n: int, rank: int, nranks: int) -> tuple[int, int, int]:
    """(lo, hi, slab) with the same rounding as fgpu_bfs_plan_create (bfs.hip)."""
    per = (n + nranks - 1) // nranks
    slab = (per + 4095) & ~4095
    return rank * slab, (rank + 1) * slab, slab


def run_levels(backend, gather: Callable[[], None], src: int, max_level: int = -1, first_batch: int = 6,
               batch: int = 3) -> int:
    """Drive one BFS: returns the number of levels.  `backend.done()` is the only host sync."""
    backend.begin(src, max_level)
    k = first_batch
    while True:
        for _ in range(k):
            backend.step()
            gather()
            backend.commit()
        done, level = backend.done()
        if done:
            return level
        k = batch


class HipSlabBackend:
    """Product backend: libfgpu plan stepping on this rank's slab, buffers owned by torch."""

    def __init__(self, ctx, A_slab, At_slab, rank: int, nranks: int, device):
        import torch
        from .engine import BfsPlan

        self.torch = torch
        self.plan = BfsPlan(ctx, A_slab, At_slab, rank, nranks)
        _, _, wpr = self.plan.part_buffers()
        self.words_per_rank = int(wpr)
        # torch owns the exchange buffers so torch.distributed can use them directly
        self.local = torch.zeros(self.words_per_rank, dtype=torch.int64, device=device)
        self.glob = torch.zeros(self.words_per_rank * nranks, dtype=torch.int64, device=device)
        self.nranks = nranks
        if nranks > 1:
            self.plan.part_set_buffers(self.local.data_ptr(), self.glob.data_ptr())
        else:
            self.plan.part_set_buffers(self.glob.data_ptr(), self.glob.data_ptr())

    def begin(self, src, max_level=-1):
        self.plan.part_begin(src, max_level)

    def step(self):
        self.plan.part_step()

    def commit(self):
        self.plan.part_commit()

    def done(self):
        return self.plan.part_done()

    def gather(self):
        if self.nranks > 1:
            self.torch.distributed.all_gather_into_tensor(self.glob, self.local)

    def run(self, src, max_level=-1):
        return run_levels(self, self.gather, src, max_level)
