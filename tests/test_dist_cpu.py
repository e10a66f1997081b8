"""CPU, world_size 2, gloo: the multi-GPU BFS control flow (falkordb_amd.dist.run_levels: column-slab
partition, one frontier all-gather per level, device-side termination) driven with an oracle-backed
stand-in for the HIP step kernels.  The HIP backend itself is covered on the GPU by
tests/test_gpu_dist.py (two slab plans on one device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import oracle
from falkordb_amd import dist as fdist

try:                                             # the partition arithmetic (slab_layout, balanced_splits) lives in libfgpu.so, which
    from falkordb_amd import _ffi                # needs the HIP runtime to LOAD (no device): a box without ROCm skips this module
    _ffi.load()
except Exception as _e:   # noqa: BLE001
    pytest.skip(f"libfgpu.so does not load here ({_e!r}): the partition functions are part of it", allow_module_level=True)

U64 = np.uint64


class OracleSlabBackend:
    """Same contract as dist.HipSlabBackend, numpy inside: rank owns destinations [lo, hi)."""

    def __init__(self, a: oracle.CSR, rank, nranks, splits=None):
        self.n = a.nrows
        self.rank, self.nranks = rank, nranks
        if splits is None:                       # equal slabs (fgpu_bfs_plan_create)
            self.lo, self.hi, self.slab = fdist.slab_range(self.n, rank, nranks)
            self.npad = self.slab * nranks
        else:                                    # caller-chosen, nnz-balanced, uneven (fgpu_bfs_plan_create_slab)
            self.lo, self.hi = int(splits[rank]), int(splits[rank + 1])
            self.slab = self.hi - self.lo
            self.npad = int(splits[nranks])
        self.splits = splits
        rows, cols = a.pairs()
        keep = (cols >= self.lo) & (cols < self.hi)
        self.a_slab = oracle.build_csr(self.n, self.n, rows[keep], cols[keep])   # A[:, lo:hi)
        self.words_per_rank = self.slab // 64
        self.local = torch.zeros(max(self.words_per_rank, 1), dtype=torch.int64)
        self.glob = torch.zeros(self.npad // 64, dtype=torch.int64)
        self.level = np.full(self.n, -1, dtype=np.int32)

    def gather(self):
        if self.splits is None:
            td.all_gather_into_tensor(self.glob, self.local)
        else:
            fdist.allgatherv_words(self.glob, self.local, self.splits, self.rank, self.nranks, td.all_gather_into_tensor)

    def begin(self, src, max_level=-1):
        npad = self.npad
        self.cur = np.zeros(npad // 64, dtype=U64)
        self.cur[src >> 6] = U64(1) << U64(src & 63)
        self.visited = self.cur.copy()
        self.level[:] = -1
        if self.lo <= src < self.hi:
            self.level[src] = 0
        self.lvl, self.max_level, self.is_done = 0, max_level, (max_level == 0)

    def step(self):
        if self.is_done:
            return
        nw = (self.n + 63) // 64
        w = oracle.vxm(self.a_slab, self.cur[:nw], self.visited[:nw])     # q<!visited> = q x A[:, slab]
        full = np.zeros(len(self.cur), dtype=U64)
        full[:nw] = w
        lo_w = self.lo // 64
        self.local[:self.words_per_rank] = torch.from_numpy(full[lo_w:lo_w + self.words_per_rank].view(np.int64).copy())

    def commit(self):
        if self.is_done:
            return
        g = self.glob.numpy().view(U64).copy()
        new_ids = oracle.ids_from_bits(g, self.n)
        self.cur = g
        self.visited = self.visited | g
        self.lvl += 1
        own = new_ids[(new_ids >= self.lo) & (new_ids < self.hi)].astype(np.int64)
        self.level[own] = self.lvl
        self.local.zero_()
        self.is_done = (len(new_ids) == 0) or (self.max_level >= 0 and self.lvl >= self.max_level)

    def done(self):
        return self.is_done, self.lvl


def _balanced(a: oracle.CSR, world):
    shift = fdist.splits_shift(a.ncols)
    counts = np.bincount((a.colidx >> np.uint64(shift)).astype(np.int64), minlength=(a.ncols + (1 << shift) - 1) >> shift)
    return fdist.balanced_splits(counts, a.ncols, world, shift)


def _worker(rank, world, port, scale, srcs, max_level, q, balanced=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = oracle.rmat_csr(scale)
        be = OracleSlabBackend(a, rank, world, _balanced(a, world) if balanced else None)
        out = []
        for src in srcs:
            nlev = fdist.run_levels(be, be.gather, src, max_level, first_batch=2, batch=1)
            lv = torch.from_numpy(be.level.copy())
            td.all_reduce(lv, op=td.ReduceOp.MAX)      # owners hold the levels, everyone else -1
            out.append((nlev, lv.numpy()))
        if rank == 0:
            q.put(out)
    finally:
        td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,max_level,balanced", [(2, -1, False), (2, 2, False), (3, -1, False),
                                                      (2, -1, True), (3, -1, True), (3, 2, True)])
def test_multi_rank_slab_bfs_matches_single_process_oracle(world, max_level, balanced):
    # world 3, equal slabs: the last one is mostly padding (slab = ceil(n / 3) rounded up to 4096).
    # balanced: nnz-balanced boundaries (the product's fgpu_mat_balanced_splits rule) — 4 blocks of 4096 vertices over
    # 3 ranks are uneven, 2 blocks over 3 ranks leave one rank an EMPTY slab; the exchange is an all-gather-v
    scale = (14 if max_level < 0 else 13) if balanced else (10 if world == 2 else 13)
    a = oracle.rmat_csr(scale)
    if balanced and world == 3:
        sp = _balanced(a, world)
        widths = {sp[r + 1] - sp[r] for r in range(world)}
        assert len(widths) > 1 and (scale != 13 or 0 in widths)
    srcs = [int(np.argmax(np.diff(a.rowptr))), 3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scale, srcs, max_level, q, balanced)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for src, (nlev, level) in zip(srcs, out):
        ref, _, _ = oracle.bfs(a, src, max_level)
        np.testing.assert_array_equal(level, ref)


def test_balanced_splits_rule():
    # boundary k = the 4096-aligned block edge whose entry prefix is nearest to k * nnz / nparts (dist.hip)
    assert fdist.balanced_splits([20, 35, 25, 20], 16384, 4) == [0, 4096, 8192, 12288, 16384]
    assert fdist.balanced_splits([70, 10, 10, 10], 16384, 2) == [0, 4096, 16384]
    assert fdist.balanced_splits([10, 10, 10, 70], 16000, 2) == [0, 12288, 16384]
    assert fdist.balanced_splits([], 100, 3) == [0, 0, 0, 4096]
    sp = fdist.balanced_splits([5] * 8192, 8192 * 8192, 8, 13)
    assert sp == [i * (1 << 23) for i in range(9)]
    assert fdist.splits_shift(1 << 22) == 12 and fdist.splits_shift(1 << 26) == 13 and fdist.splits_shift(1 << 30) == 17


def test_slab_range_matches_plan_rounding():
    # slab = ceil(n / nranks) rounded up to 4096 (bfs.hip fgpu_bfs_plan_create)
    assert fdist.slab_range(1 << 22, 0, 1) == (0, 1 << 22, 1 << 22)
    assert fdist.slab_range(1 << 22, 3, 8) == (3 * (1 << 19), 4 * (1 << 19), 1 << 19)
    lo, hi, slab = fdist.slab_range(1000, 1, 2)
    assert slab == 4096 and (lo, hi) == (4096, 8192)


# ---- k-hop MATCH: source rows sharded over ranks, layers replicated -------------------------

def _oracle_expand(a, dp, dm, hops):
    n = a.nrows

    def run(src):
        k = len(src)
        c = oracle.build_csr(k, n, np.arange(k, dtype=U64), np.asarray(src, dtype=U64))
        flops = 0
        for _ in range(hops):
            c, fl = oracle.delta_lmxm(c, a, dp, dm)
            flops += fl
        return c.rowptr.astype(U64), c.colidx.astype(U64), flops
    return run


def _expand_inputs(scale, nsrc):
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rng = np.random.default_rng(11)
    rows, cols = a.pairs()
    pick = rng.choice(len(rows), 60, replace=False)
    dm = oracle.build_csr(n, n, rows[pick], cols[pick])
    dp = oracle.build_csr(n, n, rng.integers(0, n, 60).astype(U64), rng.integers(0, n, 60).astype(U64))
    src = rng.integers(0, n, nsrc).astype(U64)
    return a, dp, dm, src


def _expand_worker(rank, world, port, scale, nsrc, hops, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, dp, dm, src = _expand_inputs(scale, nsrc)
        run = _oracle_expand(a, dp, dm, hops)
        rp, dest, flops = fdist.expand_sharded(run, src, rank, world)
        _, _, nnz_tot, flops_tot = fdist.expand_sharded(run, src, rank, world, collect=False)
        if rank == 1:   # every rank holds the whole result; check the non-zero one
            q.put((rp, dest, flops, nnz_tot, flops_tot))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("nsrc", [37, 1])
def test_two_rank_sharded_expand_matches_single_process_oracle(nsrc):
    scale, hops = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_expand_worker, args=(r, 2, port, scale, nsrc, hops, q)) for r in range(2)]
    for p in procs:
        p.start()
    rp, dest, flops, nnz_tot, flops_tot = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, dp, dm, src = _expand_inputs(scale, nsrc)
    rp_ref, dest_ref, flops_ref = _oracle_expand(a, dp, dm, hops)(src)
    np.testing.assert_array_equal(rp, rp_ref)
    np.testing.assert_array_equal(dest, dest_ref)
    assert flops == flops_ref == flops_tot and nnz_tot == len(dest_ref)


def test_shard_rows_cover_batch_in_order():
    for nrows in (0, 1, 7, 1024, 1025):
        for nranks in (1, 2, 8):
            spans = [fdist.shard_rows(nrows, r, nranks) for r in range(nranks)]
            assert spans[0][0] == 0 and spans[-1][1] == nrows
            assert all(spans[i][1] == spans[i + 1][0] for i in range(nranks - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- the partition arithmetic itself: libfgpu.so's pure host functions (no device needed) ------------------------------

def test_slab_layout_of_the_library_covers_the_bitmap_exactly():
    """fgpu_slab_layout is what fgpu_bfs_plan_create, the per-level exchange (comm_allgatherv_u64 / the peer scatter) and
    every launcher use: ranges ascend in multiples of 4096, the words tile [0, nw) with no gap or overlap — for equal
    slabs and for caller-chosen, uneven and EMPTY ones."""
    for n, nranks in [(1 << 14, 1), (1 << 14, 2), (1 << 14, 3), (100_000, 8), (4096, 4), (5, 3)]:
        lo, hi, off, cnt = fdist.slab_layout(n, nranks)
        per = -(-n // nranks)
        slab = (per + 4095) // 4096 * 4096
        assert lo == [r * slab for r in range(nranks)] and hi == [(r + 1) * slab for r in range(nranks)]
        assert off == [x // 64 for x in lo] and cnt == [slab // 64] * nranks
        assert fdist.slab_range(n, nranks - 1, nranks) == (lo[-1], hi[-1], slab)
    splits = [0, 8192, 8192, 12288, 16384]            # rank 1 owns nothing
    lo, hi, off, cnt = fdist.slab_layout(16000, 4, splits)
    assert (lo, hi) == (splits[:-1], splits[1:])
    assert cnt == [128, 0, 64, 64] and off == [0, 128, 128, 192] and sum(cnt) == 16384 // 64
    with pytest.raises(Exception):
        fdist.slab_layout(16000, 2, [0, 100, 16384])   # not a multiple of 4096


def test_balanced_splits_host_half_balances_a_skewed_histogram():
    """fgpu_balanced_splits_from_hist on a hub-heavy column histogram: every part holds its share of the entries to
    within one block, boundaries are block edges, nothing moves backwards; one part and more parts than blocks."""
    rng = np.random.default_rng(11)
    n, shift = 1 << 18, 12
    counts = (rng.pareto(1.2, n >> shift) * 1000).astype(np.uint64) + 1
    counts[:4] *= 50                                      # hubs cluster in the low ids (unscrambled R-MAT, real graphs)
    nnz = int(counts.sum())
    assert fdist.splits_shift(n) == 12 and fdist.splits_shift(1 << 26) == 13
    for nparts in (1, 2, 3, 8):
        sp = fdist.balanced_splits(counts, n, nparts, shift)
        assert sp[0] == 0 and sp[-1] == n and all(x % 4096 == 0 for x in sp) and sp == sorted(sp)
        pre = np.concatenate([[0], np.cumsum(counts)])
        per = [int(pre[b >> shift] - pre[a >> shift]) for a, b in zip(sp, sp[1:])]
        assert sum(per) == nnz
        for k in range(1, nparts):                        # each boundary is the nearest block edge to its target
            t = nnz * k / nparts
            e = sp[k] >> shift
            assert abs(int(pre[e]) - t) <= min(abs(int(pre[max(e - 1, 0)]) - t), abs(int(pre[min(e + 1, len(pre) - 1)]) - t)) + 1e-9
    sp = fdist.balanced_splits(counts[:2], 8192, 8, shift)   # 2 blocks, 8 parts: empty slabs, still a valid partition
    assert sp[0] == 0 and sp[-1] == 8192 and sp == sorted(sp) and len(sp) == 9
