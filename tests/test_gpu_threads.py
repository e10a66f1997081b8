"""The boundary under the reference's threading model: many reader threads, ONE context, shared snapshots.

The reference calls GraphBLAS from a worker pool on shared materialised handles (threadpool.rs:89-128; Matrix::wait's
double-checked lock, matrix.rs:781-796).  include/fgpu.h gives the same contract: every host thread gets its own lane
(stream + staging + free-list) inside the one fgpu_ctx, snapshots are immutable, their acceleration indexes are
built once under a per-snapshot mutex.  These tests run >= 8 host threads (ctypes releases the GIL inside every call)
through one context on the SAME snapshots — including the first-use builds of every lazy index (transpose cache,
item lists, wordrow, hub lists, hub-first order, LDS tiles) racing each other — and compare every result with the
CPU oracle."""
import queue
import threading

import numpy as np
import pytest

import oracle
from falkordb_amd import engine

pytestmark = pytest.mark.gpu
U64 = np.uint64
NTHREADS = 10


def up(ctx, a: oracle.CSR):
    return ctx.mat_from_csr(a.nrows, a.ncols, a.rowptr, a.colidx)


def run_threads(fn, n=NTHREADS):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:   # noqa: BLE001 — reported below with the thread id
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=wrap, args=(t,)) for t in range(n)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs


def test_reader_threads_share_snapshots_through_one_context(ctx):
    scale = 15
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rng = np.random.default_rng(15)
    dm_h = oracle.sample(a, 0x7EAD, 200)
    k = a.nnz // 200
    dp_h = oracle.build_csr(n, n, rng.integers(0, n, k).astype(U64), rng.integers(0, n, k).astype(U64))
    dp_h = oracle.merge(dp_h, None, a)                       # dp ∩ m = ∅
    at = oracle.transpose(a)
    # fresh snapshots: no lazy index exists yet, so the threads race to build each of them
    A, dp, dm = up(ctx, a), up(ctx, dp_h), up(ctx, dm_h)
    At = up(ctx, at)
    deg = np.diff(a.rowptr)
    roots = [int(r) for r in np.nonzero(deg > 0)[0][:NTHREADS]]
    srcs = [rng.choice(n, 96 + 8 * t, replace=False).astype(U64) for t in range(NTHREADS)]
    # oracle answers, computed up front on the main thread
    ref_expand = [oracle.expand_omp(s, [(a, dp_h, dm_h)] * 2, threads=2) for s in srcs]
    ref_expand3 = [oracle.expand_omp(s[:64], [(a, None, None)] * 3, threads=2) for s in srcs]
    ref_bfs = [oracle.bfs(a, r, -1)[0] for r in roots]
    pr = rng.integers(0, n, 5000).astype(U64)
    pc = np.where(rng.random(5000) < 0.5, rng.integers(0, n, 5000), 0).astype(U64)
    hit = np.arange(0, a.nnz, max(1, a.nnz // 2500))
    rows_of = np.repeat(np.arange(n, dtype=U64), deg.astype(np.int64))
    pr[:len(hit)] = rows_of[hit]
    pc[:len(hit)] = a.colidx[hit]
    s_ = a.to_set()
    ref_probe = np.array([(int(r), int(c)) in s_ for r, c in zip(pr, pc)], dtype=np.uint8)
    ref_merge = oracle.merge(a, dp_h, dm_h)
    fbits = oracle.bits_from_ids(n, rng.choice(n, n // 5, replace=False))
    ref_vxm = oracle.vxm(a, fbits, None)
    barrier = threading.Barrier(NTHREADS)

    def worker(t):
        barrier.wait()                                       # everyone hits the first-use builds together
        for it in range(6):
            op = (t + it) % 6
            if op == 0:                                      # CondTraverse core, dirty layers (CSR products + merge)
                c, flops, _ = ref_expand[t]
                rowptr, dest, fl = engine.expand(ctx, srcs[t], [A, A], [dp, dp], [dm, dm])
                assert fl == flops and np.array_equal(rowptr, c.rowptr) and np.array_equal(dest, c.colidx), ("expand", t)
            elif op == 1:                                    # the same chain in bit form: transpose cache + item lists
                c, flops, _ = ref_expand3[t]
                lv = engine.expand_levels(ctx, srcs[t][:64], [A] * 3)
                assert lv["hop_nnz"][-1] == c.nnz and lv["flops"] == flops, ("levels", t)
                assert lv["hop_checksum"][-1] == oracle.checksum_omp(c, 1), ("levels checksum", t)
            elif op == 2:                                    # algo.BFS on an own plan: hub lists, hub-first order
                level, _, _ = engine.bfs(ctx, A, At, roots[t], -1, want_parent=False)
                assert np.array_equal(level, ref_bfs[t]), ("bfs", t)
            elif op == 3:                                    # ExpandInto / label probes
                assert np.array_equal(A.probe(pr, pc), ref_probe), ("probe", t)
            elif op == 4:                                    # Delta fold: wordrow index
                m = A.merge(dp, dm)
                rp, ci, _ = m.export_csr()
                assert np.array_equal(rp, ref_merge.rowptr) and np.array_equal(ci, ref_merge.colidx), ("merge", t)
            else:                                            # dense-frontier vxm over the LDS tiles of A'
                w = engine.vxm(ctx, fbits, None, A, At, direction=3)
                assert np.array_equal(w, ref_vxm), ("vxm", t)

    run_threads(worker)


def test_snapshots_created_on_one_thread_are_used_and_freed_on_others(ctx):
    """Handles cross threads: producers build transposes / merges / products and hand them over the moment the call
    returns; consumers read them on their own lanes and free them there (fgpu_mat_free's cross-lane fence)."""
    a = oracle.rmat_csr(13)
    n = a.nrows
    A = up(ctx, a)
    at = oracle.transpose(a)
    rng = np.random.default_rng(13)
    q = queue.Queue(maxsize=4)
    nprod, ncons, items = 3, 5, 12

    def producer(t):
        r = np.random.default_rng(100 + t)
        for i in range(items):
            kind = (t + i) % 3
            if kind == 0:
                q.put(("transpose", A.transpose(), None))
            elif kind == 1:
                src = r.choice(n, 40, replace=False).astype(U64)
                f = ctx.mat_from_coo(40, n, np.arange(40, dtype=U64), src)
                q.put(("product", f.mxm(A), src))
            else:
                k = 300
                dp_h = oracle.build_csr(n, n, r.integers(0, n, k).astype(U64), r.integers(0, n, k).astype(U64))
                q.put(("merge", A.merge(up(ctx, dp_h), None), dp_h))

    def consumer(t):
        while True:
            item = q.get()
            if item is None:
                return
            kind, m, aux = item
            rp, ci, _ = m.export_csr()
            if kind == "transpose":
                ref = at
            elif kind == "product":
                ref, _ = oracle.mxm(oracle.build_csr(40, n, np.arange(40, dtype=U64), aux), a)
            else:
                ref = oracle.merge(a, aux, None)
            assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx), kind
            m.free()                                         # freed on a lane that did not create it

    errs = []

    def guard(fn, t):
        try:
            fn(t)
        except BaseException as e:   # noqa: BLE001
            errs.append((fn.__name__, t, repr(e)))

    prods = [threading.Thread(target=guard, args=(producer, t)) for t in range(nprod)]
    cons = [threading.Thread(target=guard, args=(consumer, t)) for t in range(ncons)]
    for th in prods + cons:
        th.start()
    for th in prods:
        th.join()
    for _ in cons:
        q.put(None)
    for th in cons:
        th.join()
    assert not errs, errs
    assert rng is not None


def test_threads_come_and_go(ctx):
    """Lanes of exited threads are handed to new threads; a context outlives many short-lived workers."""
    a = oracle.rmat_csr(11)
    A = up(ctx, a)
    ref = oracle.transpose(a)
    for wave in range(4):
        def worker(t):
            T = A.transpose()
            rp, ci, _ = T.export_csr()
            assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
        run_threads(worker, 6)
    in_use, pooled = ctx.device_bytes()
    assert in_use > 0


@pytest.mark.timeout(180)
def test_search_plans_driven_from_many_threads_at_once(ctx):
    """Six query threads, each pipelining searches over its own pair of plans (run_async / wait, the bench's pattern) on one
    shared graph.  With the CUs shared between streams a light level's control step can run before workgroups of the same
    launch that take no part have started; such a latecomer used to join the NEXT level and strand its ticket (three threads
    stalled for ever — tools/experiments/bfs_threads_hang.py).  Every search must finish and equal the oracle."""
    a = oracle.rmat_csr(18)
    A = up(ctx, a)
    At = A.transpose()
    deg = np.diff(a.rowptr)
    roots = [int(r) for r in np.nonzero(deg > 0)[0][:12]]
    ref = {r: oracle.bfs(a, r, -1)[0] for r in roots}

    def worker(t):
        plans = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
        mine = [roots[(t + i) % len(roots)] for i in range(40)]
        for i, r in enumerate(mine):
            plans[i % 2].run_async(r, -1, False, 0)
            if i > 0:
                plans[(i - 1) % 2].wait()
                if i % 7 == 0:                                   # (fetching costs a 1 MB copy: a sample, and the last two below)
                    assert np.array_equal(plans[(i - 1) % 2].fetch()[0], ref[mine[i - 1]]), ("search", t, i - 1)
        plans[(len(mine) - 1) % 2].wait()
        assert np.array_equal(plans[(len(mine) - 1) % 2].fetch()[0], ref[mine[-1]]), ("last search", t)
        assert np.array_equal(plans[(len(mine) - 2) % 2].fetch()[0], ref[mine[-2]]), ("last but one", t)
        for p in plans:
            p.free()

    run_threads(worker, 6)
