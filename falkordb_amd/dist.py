"""Multi-GPU drivers.  BFS: column-slab partition + one frontier all-gather per level.  k-hop MATCH: source
rows sharded, layers replicated, no data-path collective (expand_sharded, end of file).

SURVEY.md §8e / BASELINE.json north_star: rank r owns destination vertices
[r*slab, (r+1)*slab) and holds A[:, slab] (push) and A'[slab, :] (pull).  A column slab only ever
discovers vertices it owns, so a level is ONE kernel per rank (libfgpu fgpu_bfs_slab_level: discovery,
visited / level / parent update and the rank's owned words of the next frontier) followed by ONE
collective: the equal 1-D slabs are all-gathered over RCCL/xGMI (`all_gather_into_tensor`), and the
gathered buffer IS the global frontier bitmap the next level reads.  Termination is decided from the
population of that bitmap (identical on every rank); the push / pull choice is rank-local (either
direction yields the same owned bits), so no other collective is needed.  Levels are enqueued blind in
batches; the host looks at the device control block once per batch.

The level loop is generic over a `backend` (begin / step / commit / done) and a `gather`
callable so that tests can drive the exact same control flow on CPU with gloo
(tests/test_dist_cpu.py supplies an oracle-backed backend; the product backend below is the
HIP one and has no CPU fallback).
"""
from __future__ import annotations

from typing import Callable


def _lib():
    """libfgpu.so: the ONE definition of the partition arithmetic (fgpu_slab_layout / fgpu_balanced_splits_from_hist /
    fgpu_splits_shift, pure host functions of dist.hip).  The launchers, the bench and the CPU (gloo) tests all call it —
    there is no Python restatement beside it; a box where the library cannot be loaded cannot run the product either."""
    from . import _ffi
    return _ffi.load(), _ffi


def splits_shift(ncols: int) -> int:
    """fgpu_splits_shift: log2 of the column-block width of the balancing histogram (at most 8192 blocks)."""
    lib, _ = _lib()
    return int(lib.fgpu_splits_shift(int(ncols)))


def slab_layout(n: int, nranks: int, splits=None):
    """fgpu_slab_layout (dist.hip) — the library's own partition arithmetic, a pure host function: per rank the vertex
    range [lo, hi) and its (word offset, word count) in the global frontier bitmap.  splits=None: the equal slabs of
    fgpu_bfs_plan_create."""
    lib, ffi = _lib()
    import ctypes as C

    import numpy as np
    lo, hi, off, cnt = (np.zeros(nranks, dtype=np.uint64) for _ in range(4))
    sp = np.ascontiguousarray(splits, dtype=np.uint64) if splits is not None else None
    p = lambda x: x.ctypes.data_as(C.POINTER(C.c_uint64)) if x is not None else None
    ffi.check(lib.fgpu_slab_layout(p(sp), int(n), int(nranks), p(lo), p(hi), p(off), p(cnt)))
    return lo.tolist(), hi.tolist(), off.tolist(), cnt.tolist()


def slab_range(n: int, rank: int, nranks: int) -> tuple[int, int, int]:
    """(lo, hi, slab) of an equal-slab partition: fgpu_slab_layout with no splits (= fgpu_bfs_plan_create's rounding)."""
    lo, hi, _, _ = slab_layout(n, nranks)
    return int(lo[rank]), int(hi[rank]), int(hi[rank] - lo[rank])


def balanced_splits(block_counts, n: int, nparts: int, shift: int = 12):
    """fgpu_balanced_splits_from_hist (dist.hip): the host half of fgpu_mat_balanced_splits — `block_counts[b]` = entries
    whose column lies in block b of 2**shift columns; boundary k is the block edge whose entry prefix is nearest to
    k * nnz / nparts; boundaries are multiples of 4096, start at 0 and end at n rounded up to 4096."""
    lib, ffi = _lib()
    import ctypes as C

    import numpy as np
    h = np.ascontiguousarray(block_counts, dtype=np.uint64)
    out = np.zeros(nparts + 1, dtype=np.uint64)
    u64p = C.POINTER(C.c_uint64)
    ffi.check(lib.fgpu_balanced_splits_from_hist(h.ctypes.data_as(u64p), len(h), int(shift), int(n), int(nparts),
                                                 out.ctypes.data_as(u64p)))
    return [int(x) for x in out]


def allgatherv_words(glob, piece, splits, rank: int, nranks: int, all_gather):
    """The frontier exchange of fgpu_bfs_dist_run (dist.hip comm_allgatherv_u64) for launchers whose collective only
    takes equal pieces (gloo in the CPU tests): rank r's words live at the offset fgpu_slab_layout gives.
    `all_gather(out, inp)` = all_gather_into_tensor of equal-size tensors; pieces are padded to the widest slab."""
    import torch
    _, _, offs, words = slab_layout(int(splits[nranks]), nranks, splits)
    wmax = max(max(words), 1)
    pad = torch.zeros(wmax, dtype=glob.dtype, device=glob.device)
    pad[:words[rank]] = piece[:words[rank]]
    out = torch.zeros(wmax * nranks, dtype=glob.dtype, device=glob.device)
    all_gather(out, pad)
    for r in range(nranks):
        glob[offs[r]:offs[r] + words[r]] = out[r * wmax:r * wmax + words[r]]


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
    """Product backend, fused slab path: ONE level kernel + ONE all-gather per level (fgpu_bfs_slab_*).
    Buffers are torch tensors so torch.distributed can use them directly.  `protocol="stepped"` selects the
    older step / gather / commit kernels (kept for comparison and as a second implementation in the tests)."""

    def __init__(self, ctx, A_slab, At_slab, rank: int, nranks: int, device, protocol: str = "fused"):
        import torch
        from .engine import BfsPlan

        self.torch = torch
        self.protocol = protocol
        self.last_levels = 0
        self.plan = BfsPlan(ctx, A_slab, At_slab, rank, nranks)
        _, _, wpr = self.plan.part_buffers()
        self.words_per_rank = int(wpr)
        self.nranks = nranks
        self.glob = torch.zeros(self.words_per_rank * nranks, dtype=torch.int64, device=device)
        if protocol == "stepped":
            self.local = torch.zeros(self.words_per_rank, dtype=torch.int64, device=device)
            if nranks > 1:
                self.plan.part_set_buffers(self.local.data_ptr(), self.glob.data_ptr())
            else:
                self.plan.part_set_buffers(self.glob.data_ptr(), self.glob.data_ptr())
            return
        self.send = torch.zeros(2, self.words_per_rank, dtype=torch.int64, device=device)
        self.send_index = 0
        self.plan.slab_set_buffers(self.send[0].data_ptr(), self.send[1].data_ptr(), self.glob.data_ptr())
        # global out-degrees: every rank's column slab knows its own share of each row; the sum over ranks is
        # the degree the owner accounts at discovery (TEPS numerator) and feeds its push / pull rule with
        self.deg = torch.zeros(int(A_slab.nrows), dtype=torch.int32, device=device)
        A_slab.row_degrees(self.deg.data_ptr())
        if nranks > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(self.deg, op=torch.distributed.ReduceOp.SUM)
        self.plan.slab_set_degrees(self.deg.data_ptr())

    def set_degrees(self, deg):
        """Single-process gangs (tests): install the summed degree vector by hand."""
        self.deg = deg
        self.plan.slab_set_degrees(deg.data_ptr())

    def send_buffer(self):
        return self.local if self.protocol == "stepped" else self.send[self.send_index]

    def begin(self, src, max_level=-1):
        if self.protocol == "stepped":
            self.plan.part_begin(src, max_level)
        else:
            self.plan.slab_begin(src, max_level, False)

    def step(self):
        if self.protocol == "stepped":
            self.plan.part_step()
        else:
            self.send_index = self.plan.slab_level()

    def commit(self):
        if self.protocol == "stepped":
            self.plan.part_commit()

    def done(self):
        return self.plan.part_done()

    def gather(self):
        if self.nranks > 1:
            self.torch.distributed.all_gather_into_tensor(self.glob, self.send_buffer())
        elif self.protocol != "stepped":
            self.glob.copy_(self.send_buffer())   # one rank: the "gather" is a device copy

    def run(self, src, max_level=-1):
        # first blind batch = what the previous search needed (+1); R-MAT roots differ by a level at most
        first = max(4, self.last_levels + 1) if self.last_levels else 6
        self.last_levels = run_levels(self, self.gather, src, max_level, first_batch=first, batch=2)
        return self.last_levels


# ---------------------------------------------------------------------------------------------------
# k-hop MATCH across ranks (last bullet): the source rows of F are independent units, the
# layers are replicated (RMAT-26 is 4.3 GB of 288), so a batch is split by rows, every rank runs the whole
# chain on its share with NO data-path collective, and only the results travel (to the rank that talks to
# the client).  `expand_local(src_ids) -> (rowptr, dest, flops)` is engine.expand bound to this rank's
# context and layer handles; the CPU tests bind the oracle instead.

def shard_rows(nrows: int, rank: int, nranks: int) -> tuple[int, int]:
    """Contiguous, sizes differing by at most one; rank order = row order, so concatenation keeps the
    ascending (row, dest) order CondTraverseOp::expand_batch iterates in (cond_traverse.rs:612-640)."""
    base, extra = divmod(nrows, nranks)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _gather_varlen(t, group=None):
    """All ranks' 1-D int64 tensors, concatenated in rank order (RCCL has no all-gather-v: sizes first, then one
    padded all_gather_into_tensor)."""
    import torch
    import torch.distributed as td

    world = td.get_world_size(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=t.device)
    mine = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    td.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.tolist()
    cap = max(max(sizes), 1)
    pad = torch.zeros(cap, dtype=torch.int64, device=t.device)
    pad[:t.numel()] = t
    allb = torch.empty(world * cap, dtype=torch.int64, device=t.device)
    td.all_gather_into_tensor(allb, pad, group=group)
    return torch.cat([allb[r * cap:r * cap + sizes[r]] for r in range(world)])


def expand_sharded(expand_local, src_ids, rank: int, nranks: int, device="cpu", collect: bool = True):
    """One CondTraverse batch over `nranks` GPUs.  Returns (rowptr[nsrc+1], dest, flops) of the WHOLE batch on
    every rank when `collect`, else this rank's share plus the batch totals (nnz, flops) — the
    `RETURN count(c)` shape, where only two integers cross xGMI."""
    import numpy as np
    import torch
    import torch.distributed as td

    src_ids = np.ascontiguousarray(src_ids, dtype=np.uint64)
    lo, hi = shard_rows(len(src_ids), rank, nranks)
    if hi > lo:
        rowptr, dest, flops = expand_local(src_ids[lo:hi])
    else:       # more ranks than rows: this rank only takes part in the collectives
        rowptr, dest, flops = np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint64), 0
    rowptr = np.asarray(rowptr, dtype=np.uint64)
    dest = np.asarray(dest, dtype=np.uint64)
    if nranks == 1:
        return (rowptr, dest, int(flops)) if collect else (rowptr, dest, int(len(dest)), int(flops))
    tot = torch.tensor([len(dest), int(flops)], dtype=torch.int64, device=device)
    td.all_reduce(tot, op=td.ReduceOp.SUM)
    if not collect:
        return rowptr, dest, int(tot[0]), int(tot[1])
    counts = torch.from_numpy(np.diff(rowptr.astype(np.int64))).to(device)
    counts = _gather_varlen(counts).cpu().numpy()
    dest_all = _gather_varlen(torch.from_numpy(dest.view(np.int64)).to(device)).cpu().numpy().view(np.uint64)
    rp = np.zeros(len(src_ids) + 1, dtype=np.uint64)
    np.cumsum(counts, out=rp[1:].view(np.int64))
    return rp, dest_all, int(tot[1])


def expand_count_sharded(count_local, src_ids, rank: int, nranks: int, device="cpu"):
    """`RETURN count(c)` over ranks: `count_local(src_ids) -> (nnz, flops)` is engine.expand_count on this rank's
    share (nothing is materialised); the batch totals are one 16-byte all-reduce."""
    import numpy as np
    import torch
    import torch.distributed as td

    src_ids = np.ascontiguousarray(src_ids, dtype=np.uint64)
    lo, hi = shard_rows(len(src_ids), rank, nranks)
    nnz, flops = count_local(src_ids[lo:hi]) if hi > lo else (0, 0)
    if nranks == 1:
        return int(nnz), int(flops)
    tot = torch.tensor([int(nnz), int(flops)], dtype=torch.int64, device=device)
    td.all_reduce(tot, op=td.ReduceOp.SUM)
    return int(tot[0]), int(tot[1])
