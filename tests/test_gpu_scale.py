"""Parity at (or near) the BASELINE.json sizes — the HIP path against the CPU oracle, not against itself.

The serial oracle (oracle/oracle.c) needs minutes at these sizes, so the checker here is its row-parallel twin
(oracle/oracle_omp.c: orc_mxm_omp / orc_merge_omp, held equal to the serial restatement by
tests/test_oracle_golden.py::test_openmp_products_match_the_serial_oracle).  Inputs: the R-MAT graph is generated
on the device (fgpu_mat_rmat — held equal to oracle.rmat_csr by smoke() and tests/test_gpu_matrix.py) and exported
to the host for the oracle; delta layers come from fgpu_mat_sample (uniformly random tombstones) and a seeded
host RNG (pending adds).  SURVEY.md §8d configs 3, 4, 5."""
import numpy as np
import pytest

import oracle
from falkordb_amd import engine

pytestmark = pytest.mark.gpu
U64 = np.uint64


def host_csr(A) -> oracle.CSR:
    rp, ci, _ = A.export_csr()
    return oracle.CSR(A.nrows, A.ncols, rp, ci)


def p_sources(n, count, first=0):
    """`count` vertices of the synthetic label :P of SURVEY.md §8d (mix64(id) % 16 == 0), ascending from `first`."""
    ids = np.arange(first, min(n, first + 64 * count + 4096), dtype=U64)
    ids = ids[oracle.mix64(ids) % U64(16) == 0]
    assert len(ids) >= count
    return ids[:count]


def delta_layers(ctx, A, seed, denom=1000):
    """dm = a uniform 1/denom sample of A's entries; dp = as many random coordinates outside A."""
    n = A.nrows
    dm = A.sample(seed, denom)
    rng = np.random.default_rng(seed)
    k = max(1, A.nvals // denom)
    raw = ctx.mat_from_coo(n, n, rng.integers(0, n, k, dtype=np.uint64), rng.integers(0, n, k, dtype=np.uint64))
    dp = raw.merge(None, A)          # (raw \ A): the Delta invariant dp ∩ m = ∅ (versioned_matrix.rs:214-235)
    return dp, dm


@pytest.fixture(scope="module")
def rmat20(ctx):
    A = ctx.mat_rmat(20)
    dp, dm = delta_layers(ctx, A, 0x20D)
    a = host_csr(A)
    hdp, hdm = host_csr(dp), host_csr(dm)
    # the sample rule is a function of the coordinate: the oracle draws the same tombstones from its own copy
    ref_dm = oracle.sample(a, 0x20D, 1000)
    assert np.array_equal(ref_dm.rowptr, hdm.rowptr) and np.array_equal(ref_dm.colidx, hdm.colidx)
    assert 0.0005 * a.nnz < hdm.nnz < 0.002 * a.nnz and hdp.nnz > 0
    return A, dp, dm, a, hdp, hdm


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_khop_rmat20_dirty_layers_match_the_oracle(ctx, rmat20, mode):
    """BASELINE config 3's shape one notch down (RMAT-20, 3 hops, 0.1 % uniformly random tombstones + 0.1 % pending
    adds on every hop): nnz, order-independent checksum and traversed-edge count of fgpu_expand_count against the
    oracle's delta_lmxm chain, in each expand_mode (auto / sorted-CSR products / bit-parallel)."""
    A, dp, dm, a, hdp, hdm = rmat20
    src = p_sources(a.nrows, 160)
    c, flops, hop_nnz = oracle.expand_omp(src, [(a, hdp, hdm)] * 3)
    ref = (c.nnz, oracle.checksum_omp(c), flops)
    try:
        ctx.set_option("expand_mode", mode)
        got = engine.expand_count(ctx, src, [A] * 3, [dp] * 3, [dm] * 3)
    finally:
        ctx.set_option("expand_mode", 0)
    assert got == ref, (mode, got, ref)
    assert ref[0] > 20_000_000            # the case is not trivially small


def test_khop_rmat20_full_rows_match_the_oracle(ctx, rmat20):
    """The same chain with the whole (row, dest) result compared entry by entry (fgpu_expand, what the operator emits),
    clean and dirty layers, with a destination-label bitmap on the dirty run."""
    A, dp, dm, a, hdp, hdm = rmat20
    src = p_sources(a.nrows, 24, first=5000)
    c, flops, _ = oracle.expand_omp(src, [(a, None, None)] * 3)
    rowptr, dest, fl = engine.expand(ctx, src, [A] * 3)
    assert fl == flops and np.array_equal(rowptr, c.rowptr) and np.array_equal(dest, c.colidx)
    c, flops, _ = oracle.expand_omp(src, [(a, hdp, hdm)] * 3)
    label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(3) != 0
    bits = oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])
    rowptr, dest, fl = engine.expand(ctx, src, [A] * 3, [dp] * 3, [dm] * 3, dst_label_bitmap=bits)
    keep = label[c.colidx.astype(np.int64)]
    rows = np.repeat(np.arange(c.nrows), np.diff(c.rowptr).astype(np.int64))[keep]
    assert fl == flops
    assert np.array_equal(dest, c.colidx[keep])
    assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=c.nrows))]).astype(U64))


def test_khop_rmat24_clean_matches_the_oracle(ctx):
    """BASELINE config 3 at full size (RMAT-24, 268 M edges, 3 hops, clean layers): 64 sources of the :P set —
    nnz, checksum and flops of fgpu_expand_count against the oracle, plus the per-hop sizes through
    fgpu_expand_levels."""
    A = ctx.mat_rmat(24)
    a = host_csr(A)
    src = p_sources(a.nrows, 64)
    c, flops, hop_nnz = oracle.expand_omp(src, [(a, None, None)] * 3)
    ref = (c.nnz, oracle.checksum_omp(c), flops)
    del c
    got = engine.expand_count(ctx, src, [A] * 3)
    assert got == ref, (got, ref)
    lv = engine.expand_levels(ctx, src, [A] * 3)
    assert list(lv["hop_nnz"]) == hop_nnz and lv["flops"] == flops
    assert ref[0] > 50_000_000


def test_varlen_reach_config5_standin_matches_the_oracle(ctx):
    """BASELINE config 5's stand-in (≈0.5 M vertices, ≈20 M edges: R-MAT scale 19, edge factor 38), `[*1..4]` with
    0.1 % tombstones + pending adds: per-hop set sizes / checksums and the DISTINCT union of fgpu_expand_levels
    against the oracle's iterated delta_lmxm and a row-wise union."""
    A = ctx.mat_rmat(19, 38)
    dp, dm = delta_layers(ctx, A, 0x519)
    a, hdp, hdm = host_csr(A), host_csr(dp), host_csr(dm)
    assert 15_000_000 < a.nnz < 21_000_000
    src = p_sources(a.nrows, 96)
    f = oracle.build_csr(len(src), a.nrows, np.arange(len(src), dtype=U64), src)
    hop_nnz, hop_cs, flops, union = [], [], 0, None
    for _ in range(4):
        f, fl = oracle.delta_lmxm_omp(f, a, hdp, hdm)
        flops += fl
        hop_nnz.append(f.nnz)
        hop_cs.append(oracle.checksum_omp(f))
        union = f if union is None else oracle.merge_omp(union, f, None)
    got = engine.expand_levels(ctx, src, [A] * 4, [dp] * 4, [dm] * 4)
    assert list(got["hop_nnz"]) == hop_nnz
    assert list(got["hop_checksum"]) == hop_cs
    assert got["flops"] == flops
    assert got["union_nnz"] == union.nnz and got["union_checksum"] == oracle.checksum_omp(union)


def test_bfs_rmat26_levels_are_the_bfs_levels(ctx):
    """BASELINE config 4's graph (RMAT-26, 1.06 G edges) on one device: two roots through the plan API, checked by the
    properties that pin a BFS level vector uniquely — level[src] = 0; no edge (u, v) with u reached skips a level
    (level[v] <= level[u] + 1, v reached); every reached v != src has a parent one level up joined to it by a stored
    edge; reached / edges_traversed agree with the level vector.  The edges are streamed back in row windows so the
    host never holds more than ~3 GB."""
    A = ctx.mat_rmat(26)
    At = A.transpose()
    n = A.nrows
    plan = engine.BfsPlan(ctx, A, At)
    first = A.extract(0, 4095)[0]
    roots = [int(first[0]), int(first[-1])]
    results = []
    for src in roots:
        plan.run(src, -1, want_parent=True)
        level, parent = plan.fetch(want_parent=True)
        st = plan.stats()
        assert level[src] == 0 and parent[src] == src
        assert st["reached"] == int((level >= 0).sum())
        results.append((src, level, parent, st, 0))
    deg_sum = [0, 0]
    step = 1 << 21
    for lo in range(0, n, step):
        rows, cols, _ = A.extract(lo, min(n, lo + step) - 1)
        rows = rows.astype(np.int64)
        cols = cols.astype(np.int64)
        for k, (src, level, parent, st, _) in enumerate(results):
            lu, lv = level[rows], level[cols]
            live = lu >= 0
            assert (lv[live] >= 0).all() and (lv[live] <= lu[live] + 1).all()
            deg_sum[k] += int(live.sum())
            # parent edges whose parent row lies in this window: (parent[v], v) must be a stored edge
            if lo == 0:
                child = np.nonzero((level > 0) & (parent >= 0) & (parent < step))[0][:4000]
                key = rows * n + cols
                assert np.isin(parent[child] * n + child, key).all()
    for k, (src, level, parent, st, _) in enumerate(results):
        assert st["edges_traversed"] == deg_sum[k]
        others = level > 0
        assert (level[parent[others]] + 1 == level[others]).all()
        assert (parent[level < 0] == -1).all()
        assert int((level >= 0).sum()) > n // 3
