"""GPU parity: matrix builders / transposes / probes / merges vs the CPU oracle (bit-exact).

Each test names the reference behaviour it pins (file:line relative to /root/reference).
All calls go through the C ABI (include/fgpu.h)."""
import numpy as np
import pytest

import oracle
from falkordb_amd import engine

pytestmark = pytest.mark.gpu
U64 = np.uint64


def rand_coo(rng, nrows, ncols, n):
    return rng.integers(0, nrows, n, dtype=np.uint64), rng.integers(0, ncols, n, dtype=np.uint64)


def assert_same(mat, ref: oracle.CSR):
    rp, ci, _ = mat.export_csr()
    assert mat.nrows == ref.nrows and mat.ncols == ref.ncols
    assert mat.nvals == ref.nnz
    np.testing.assert_array_equal(rp, ref.rowptr)
    np.testing.assert_array_equal(ci, ref.colidx)


def test_build_collapses_duplicates(ctx):
    # matrix.rs:1686-1695: duplicate coordinates collapse to one entry
    m = ctx.mat_from_coo(4, 4, [0, 0, 0, 3, 3], [1, 1, 2, 0, 0])
    assert m.nvals == 3
    r, c, _ = m.extract()
    assert list(zip(r.tolist(), c.tolist())) == [(0, 1), (0, 2), (3, 0)]


@pytest.mark.parametrize("nrows,ncols,n", [(1, 1, 1), (7, 5, 0), (64, 64, 500), (1000, 3000, 20000),
                                           (100000, 100000, 300000)])
def test_from_coo_host_path(ctx, nrows, ncols, n):
    rng = np.random.default_rng(nrows * 31 + n)
    r, c = rand_coo(rng, nrows, ncols, n)
    assert_same(ctx.mat_from_coo(nrows, ncols, r, c), oracle.build_csr(nrows, ncols, r, c))


def test_from_coo_device_path_skewed(ctx):
    # >= 2^20 tuples go through the device histogram / segsort path, incl. a hub row (bitmap class)
    rng = np.random.default_rng(7)
    n = (1 << 21) + 12345
    nrows = ncols = 50000
    r, c = rand_coo(rng, nrows, ncols, n)
    r[:200000] = 17          # hub row with > 4096 unique columns
    r[200000:200300] = 18    # LDS class (65..4096)
    assert_same(ctx.mat_from_coo(nrows, ncols, r, c), oracle.build_csr(nrows, ncols, r, c))


def test_hypersparse_delta_layer(ctx):
    # delta layers are pinned hypersparse (versioned_matrix.rs Delta::new); few rows of a huge matrix
    n = 1 << 22
    r = np.array([5, 5, 4000000, 123456], dtype=U64)
    c = np.array([9, 3, 1, 77], dtype=U64)
    m = ctx.mat_from_coo(n, n, r, c)
    assert m.nvals == 4
    rows, cols, _ = m.extract()
    assert list(zip(rows.tolist(), cols.tolist())) == [(5, 3), (5, 9), (123456, 77), (4000000, 1)]
    rows, cols, _ = m.extract(6, 123456)
    assert list(zip(rows.tolist(), cols.tolist())) == [(123456, 77)]
    p = m.probe([5, 5, 6, 4000000], [3, 4, 3, 1])
    assert p.tolist() == [1, 0, 0, 1]


def test_u64_build_and_probe_values(ctx):
    # Matrix::<u64>::build (matrix.rs:1186-1210) + extractElement_UINT64 (:1158-1172)
    m = ctx.mat_from_coo(10, 10, [1, 1, 2, 1], [3, 4, 0, 3], vals=[100, 200, 300, 111])
    assert m.nvals == 3
    p, v = m.probe([1, 1, 2, 0], [3, 4, 0, 0], want_vals=True)
    assert p.tolist() == [1, 1, 1, 0]
    assert v[1] == 200 and v[2] == 300 and v[0] in (100, 111)


@pytest.mark.parametrize("seed", [1, 2])
def test_transpose(ctx, seed):
    rng = np.random.default_rng(seed)
    r, c = rand_coo(rng, 3000, 2000, 40000)
    a = oracle.build_csr(3000, 2000, r, c)
    assert_same(ctx.mat_from_coo(3000, 2000, r, c).transpose(), oracle.transpose(a))


@pytest.mark.parametrize("nrows,ncols,n", [(3000, 2000, 40000), (257, 70000, 100000), (70000, 255, 90000),
                                           (1 << 17, 1 << 17, 1 << 21), (5000, 1 << 20, 300000), (40, 40, 9000)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_counting_builders_match_the_oracle(ctx, nrows, ncols, n, mode):
    """transpose.hip: the sort-free builders (stable counting sorts: the two-level form, transpose_mode 3, and the
    LDS-staged levels, transpose_mode 2 — one, two and three digits over these key spaces) against the oracle and
    against the round-1 sorter path (transpose_mode 1) — rectangular shapes, key spaces that are not a multiple of the
    bucket width or a power of two, one bucket only, heavy duplication (40 x 40 with 9000 tuples), a hub column and a
    hub row."""
    rng = np.random.default_rng(nrows + 3 * ncols + n)
    r, c = rand_coo(rng, nrows, ncols, n)
    r[: n // 8] = nrows // 3                      # hub row
    c[n // 8: n // 4] = ncols - 1                 # hub column, last key of the last bucket
    ref = oracle.build_csr(nrows, ncols, r, c)
    try:
        ctx.set_option("transpose_mode", mode)
        A = ctx.mat_from_coo(nrows, ncols, r, c)
        assert_same(A, ref)
        At = A.transpose()
        assert_same(At, oracle.transpose(ref))
        assert_same(At.transpose(), ref)
    finally:
        ctx.set_option("transpose_mode", 0)


def test_from_csr_roundtrip_and_validation(ctx):
    rng = np.random.default_rng(3)
    r, c = rand_coo(rng, 500, 500, 5000)
    a = oracle.build_csr(500, 500, r, c)
    assert_same(ctx.mat_from_csr(500, 500, a.rowptr, a.colidx), a)
    bad = a.colidx.copy()
    if len(bad) > 1:
        row0 = int(np.argmax(np.diff(a.rowptr) >= 2))
        s = int(a.rowptr[row0])
        bad[s], bad[s + 1] = bad[s + 1], bad[s]
        with pytest.raises(engine.FgpuError) as e:
            ctx.mat_from_csr(500, 500, a.rowptr, bad)
        assert e.value.code == -3


def test_probe_random(ctx):
    rng = np.random.default_rng(11)
    r, c = rand_coo(rng, 2000, 2000, 30000)
    a = oracle.build_csr(2000, 2000, r, c)
    m = ctx.mat_from_coo(2000, 2000, r, c)
    pr, pc = rand_coo(rng, 2000, 2000, 5000)
    pr[:1000], pc[:1000] = r[:1000], c[:1000]
    s = a.to_set()
    want = np.array([(int(x), int(y)) in s for x, y in zip(pr, pc)], dtype=np.uint8)
    np.testing.assert_array_equal(m.probe(pr, pc), want)


@pytest.mark.parametrize("masks_dp", [False, True])
def test_merge_delta_layers(ctx, masks_dp):
    # (m \ dm) U dp: VersionedMatrix::extract / flush (versioned_matrix.rs:609-620, 892-938)
    rng = np.random.default_rng(5)
    n = 4000
    r, c = rand_coo(rng, n, n, 60000)
    pr, pc = rand_coo(rng, n, n, 900)
    k = 700
    dr = np.concatenate([r[:k], pr[:50]])
    dc = np.concatenate([c[:k], pc[:50]])
    m, dp, dm = (oracle.build_csr(n, n, r, c), oracle.build_csr(n, n, pr, pc), oracle.build_csr(n, n, dr, dc))
    got = ctx.mat_from_coo(n, n, r, c).merge(ctx.mat_from_coo(n, n, pr, pc), ctx.mat_from_coo(n, n, dr, dc),
                                             dm_masks_dp=masks_dp)
    assert_same(got, oracle.merge(m, dp, dm, masks_dp))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_merge_rmat_hub_rows_both_kernels(ctx, mode):
    # the entry-parallel merge (merge.hip; 0 = keep bits of the base layer cleared from the delta side, 2 = every base
    # entry of a touched row searching the deltas) and the wavefront-per-row one agree with the oracle on a skewed graph
    a = oracle.rmat_csr(14)
    n = a.nrows
    rng = np.random.default_rng(9)
    ar, ac = a.pairs()
    kill = rng.choice(a.nnz, a.nnz // 50, replace=False)
    hub = int(np.argmax(np.diff(a.rowptr.astype(np.int64))))
    hub_cols = a.row(hub)
    dr = np.concatenate([ar[kill], np.full(len(hub_cols) // 2, hub, dtype=np.uint64)])
    dc = np.concatenate([ac[kill], hub_cols[::2][:len(hub_cols) // 2]])
    pr, pc = rand_coo(rng, n, n, 5000)
    pr = np.concatenate([pr, np.full(3000, hub, dtype=np.uint64)])
    pc = np.concatenate([pc, rng.integers(0, n, 3000, dtype=np.uint64)])
    ctx.set_option("merge_mode", mode)
    try:
        # merge_items: the scatter as a shifted copy by 2048-entry items with the dp insertion positions as events (the
        # default; the 3000 insertions into the hub row are its dense-events branch) / the per-word, per-entry scatter
        for items in ((1, 0) if mode == 0 else (1,)):
            ctx.set_option("merge_items", items)
            for masks_dp in (False, True):
                got = ctx.mat_rmat(14).merge(ctx.mat_from_coo(n, n, pr, pc), ctx.mat_from_coo(n, n, dr, dc),
                                             dm_masks_dp=masks_dp)
                assert_same(got, oracle.merge(a, oracle.build_csr(n, n, pr, pc), oracle.build_csr(n, n, dr, dc), masks_dp))
            assert_same(ctx.mat_rmat(14).merge(None, None), a)
            assert_same(ctx.mat_new(n, n).merge(ctx.mat_from_coo(n, n, pr, pc), None), oracle.build_csr(n, n, pr, pc))
            # rows appended behind the base's last row (a grown Delta layer): they land behind every base entry
            grown = ctx.mat_rmat(14).merge(ctx.mat_from_coo(n, n, np.array([n - 1, n - 1], dtype=np.uint64),
                                                            np.array([0, n - 1], dtype=np.uint64)), None)
            assert_same(grown, oracle.merge(a, oracle.build_csr(n, n, np.array([n - 1, n - 1], dtype=np.uint64),
                                                                np.array([0, n - 1], dtype=np.uint64)), None))
    finally:
        ctx.set_option("merge_mode", 0)
        ctx.set_option("merge_items", 1)


@pytest.mark.parametrize("n_tuples", [900, 40000])
def test_merge_u64_layers_dp_value_wins(ctx, n_tuples):
    # Tensor::flush (tensor.rs:741-797): new_m<!dm> = m (+) dp with GrB_SECOND_UINT64 — on device, both the
    # short-list and the device COO builder (>= 4096 tuples) feeding it
    rng = np.random.default_rng(n_tuples)
    n = 3000
    r, c = rand_coo(rng, n, n, n_tuples)
    v = rng.integers(0, 1 << 63, n_tuples, dtype=np.uint64)
    pr = np.concatenate([r[:n_tuples // 10], rng.integers(0, n, n_tuples // 5, dtype=np.uint64)])
    pc = np.concatenate([c[:n_tuples // 10], rng.integers(0, n, n_tuples // 5, dtype=np.uint64)])
    pv = rng.integers(0, 1 << 63, len(pr), dtype=np.uint64)
    dr, dc = r[n_tuples // 20:n_tuples // 4], c[n_tuples // 20:n_tuples // 4]
    base, add = {}, {}
    for a, b, x in zip(r.tolist(), c.tolist(), v.tolist()):
        base[(a, b)] = x
    for a, b, x in zip(pr.tolist(), pc.tolist(), pv.tolist()):
        add[(a, b)] = x
    dead = set(zip(dr.tolist(), dc.tolist()))
    M, DP, DM = ctx.mat_from_coo(n, n, r, c, v), ctx.mat_from_coo(n, n, pr, pc, pv), ctx.mat_from_coo(n, n, dr, dc)
    for masks_dp in (False, True):
        want = {k: x for k, x in base.items() if k not in dead}
        want.update({k: x for k, x in add.items() if not (masks_dp and k in dead)})
        rows, cols, vals = M.merge(DP, DM, dm_masks_dp=masks_dp).extract()
        assert vals is not None
        got = dict(zip(zip(rows.tolist(), cols.tolist()), vals.tolist()))
        assert got == want
        assert list(zip(rows.tolist(), cols.tolist())) == sorted(want)
        # the same merge with the values dropped (Tensor::extract, tensor.rs:838-850)
        prow, pcol, pval = M.merge_pattern(DP, DM, dm_masks_dp=masks_dp).extract()
        assert pval is None and list(zip(prow.tolist(), pcol.tolist())) == sorted(want)
    # a BOOL layer merged into a valued one carries the iso value 1
    rows, cols, vals = M.merge(ctx.mat_from_coo(n, n, [7], [9]), None).extract()
    got = dict(zip(zip(rows.tolist(), cols.tolist()), vals.tolist()))
    assert got[(7, 9)] == 1


def test_u64_transpose_on_device(ctx):
    rng = np.random.default_rng(77)
    n = 5000
    r, c = rand_coo(rng, n, 300, 50000)
    v = rng.integers(0, 1 << 63, 50000, dtype=np.uint64)
    want = {}
    for a, b, x in zip(r.tolist(), c.tolist(), v.tolist()):
        want[(b, a)] = x
    rows, cols, vals = ctx.mat_from_coo(n, 300, r, c, v).transpose().extract()
    assert dict(zip(zip(rows.tolist(), cols.tolist()), vals.tolist())) == want
    assert list(zip(rows.tolist(), cols.tolist())) == sorted(want)


def test_resize_grow_and_shrink(ctx):
    # GrB_Matrix_resize: grow keeps every (i, j, v) (matrix.rs:1617-1672), shrink drops what falls outside
    a = oracle.rmat_csr(12)
    A = ctx.mat_rmat(12)
    G = A.resize(a.nrows + 1000, a.ncols + 77)
    assert (G.nrows, G.ncols) == (a.nrows + 1000, a.ncols + 77)
    rows, cols, _ = G.extract()
    ar, ac = a.pairs()
    np.testing.assert_array_equal(rows, ar)
    np.testing.assert_array_equal(cols, ac)
    S = A.resize(1500, 2200)
    keep = (ar < 1500) & (ac < 2200)
    assert_same(S, oracle.build_csr(1500, 2200, ar[keep], ac[keep]))
    # hypersparse delta layer grows too
    H = ctx.mat_from_coo(1 << 20, 1 << 20, [5, 900000], [6, 7]).resize((1 << 20) + 5, 1 << 21)
    rows, cols, _ = H.extract()
    assert list(zip(rows.tolist(), cols.tolist())) == [(5, 6), (900000, 7)]


def test_intersect(ctx):
    rng = np.random.default_rng(6)
    n = 3000
    r, c = rand_coo(rng, n, n, 50000)
    r2 = np.concatenate([r[:10000], rng.integers(0, n, 10000, dtype=np.uint64)])
    c2 = np.concatenate([c[:10000], rng.integers(0, n, 10000, dtype=np.uint64)])
    sa, sb = oracle.build_csr(n, n, r, c).to_set(), oracle.build_csr(n, n, r2, c2).to_set()
    inter = sorted(sa & sb)
    A, B = ctx.mat_from_coo(n, n, r, c), ctx.mat_from_coo(n, n, r2, c2)
    assert A.intersect_nvals(B) == len(inter)
    rows, cols, _ = A.intersect(B).extract()
    assert list(zip(rows.tolist(), cols.tolist())) == inter


@pytest.mark.parametrize("scale", [10, 14])
def test_rmat_matches_numpy_generator(ctx, scale):
    assert_same(ctx.mat_rmat(scale), oracle.rmat_csr(scale))


def test_slabs(ctx):
    a = oracle.rmat_csr(12)
    A = ctx.mat_rmat(12)
    lo, hi = 1024, 3072
    rp, ci, _ = A.col_slab(lo, hi).export_csr()
    rows, cols = a.pairs()
    keep = (cols >= lo) & (cols < hi)
    ref = oracle.build_csr(a.nrows, a.ncols, rows[keep], cols[keep])
    np.testing.assert_array_equal(rp, ref.rowptr)
    np.testing.assert_array_equal(ci, ref.colidx)
    rp, ci, _ = A.row_slab(lo, hi).export_csr()
    keep = (rows >= lo) & (rows < hi)
    ref = oracle.build_csr(a.nrows, a.ncols, rows[keep], cols[keep])
    np.testing.assert_array_equal(rp, ref.rowptr)
    np.testing.assert_array_equal(ci, ref.colidx)


@pytest.mark.parametrize("hyper", [False, True], ids=["dense-rowptr", "hypersparse-m"])
def test_merge_clean_word_path_with_sparse_deltas(ctx, hyper):
    """The clean-word fast path of merge.hip (a 64-entry word none of whose rows a delta touches is a shifted copy):
    a large base with long runs of EMPTY rows, dp inserting into empty rows that sit between the rows of one word
    (the word must then take the per-entry path: its offset changes mid-word), dm removing single entries, a few
    dirty rows per thousand so that clean and dirty words interleave; also the no-delta copy, a grow and a shrink
    (clip) resize, and UINT64 values riding along."""
    rng = np.random.default_rng(77)
    n = 200_000 if not hyper else 3_000_000
    live = np.sort(rng.choice(n, 30_000, replace=False))             # rows that store anything
    deg = rng.integers(1, 9, len(live))
    deg[rng.choice(len(live), 40, replace=False)] = 700              # some rows spanning many words
    rows = np.repeat(live, deg).astype(np.uint64)
    cols = rng.integers(0, n, len(rows), dtype=np.uint64)
    m = oracle.build_csr(n, n, rows, cols)
    mr, mc = m.pairs()
    # dp: 300 entries into rows that are EMPTY in m (between live rows) + 200 into live rows
    empty_rows = np.setdiff1d(rng.choice(n, 2000, replace=False), live)[:300]
    pr = np.concatenate([empty_rows, rng.choice(live, 200)]).astype(np.uint64)
    pc = rng.integers(0, n, len(pr), dtype=np.uint64)
    kill = rng.choice(m.nnz, 400, replace=False)
    dr, dc = mr[kill], mc[kill]
    dp, dm = oracle.build_csr(n, n, pr, pc), oracle.build_csr(n, n, dr, dc)
    M = ctx.mat_from_coo(n, n, mr, mc)                                # device COO build
    if hyper:
        assert M.export_csr()[0] is not None
    DP, DM = ctx.mat_from_coo(n, n, pr, pc), ctx.mat_from_coo(n, n, dr, dc)
    for masks_dp in (False, True):
        assert_same(M.merge(DP, DM, dm_masks_dp=masks_dp), oracle.merge(m, dp, dm, masks_dp))
    assert_same(M.merge(DP, None), oracle.merge(m, dp, None))
    assert_same(M.merge(None, DM), oracle.merge(m, None, dm))
    assert_same(M.merge(None, None), m)
    # resize: growing keeps every entry (no clip: fast path), shrinking drops rows / columns past the new dims
    big = M.resize(n + 1000, n + 5)
    assert big.nrows == n + 1000 and big.nvals == m.nnz
    keep = (mr < n // 2) & (mc < n // 3)
    assert_same(M.resize(n // 2, n // 3), oracle.build_csr(n // 2, n // 3, mr[keep], mc[keep]))
    # UINT64 layers: values follow their entries through the shifted copy
    vals = (mr * np.uint64(1_000_003) + mc).astype(np.uint64)
    V = ctx.mat_from_coo(n, n, mr, mc, vals)
    r2, c2, v2 = V.merge(DP, DM).extract()
    ref = oracle.merge(m, dp, dm)
    rr, rc = ref.pairs()
    np.testing.assert_array_equal(r2, rr)
    np.testing.assert_array_equal(c2, rc)
    from_m = ~np.isin(rr * np.uint64(n) + rc, pr * np.uint64(n) + pc)
    np.testing.assert_array_equal(v2[from_m], (rr * np.uint64(1_000_003) + rc)[from_m])
    assert (v2[~from_m] == 1).all()                                   # BOOL dp entries carry the iso value 1


def test_merge_random_configurations(ctx):
    """Seeded fuzz of the entry-parallel merge (clean-word and per-entry paths interleaved): 40 random shapes — row
    distributions from dense to mostly empty, deltas from none to heavy, dp landing in empty rows, dm hitting rows
    that span several 64-entry words — each against the oracle, both masking modes."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.choice([70, 500, 3000, 20000]))
        fill = float(rng.choice([0.02, 0.2, 0.9]))
        live = np.nonzero(rng.random(n) < fill)[0]
        if len(live) == 0:
            live = np.array([0])
        deg = rng.integers(1, int(rng.choice([3, 12, 200])) + 1, len(live))
        rows = np.repeat(live, deg).astype(np.uint64)
        cols = rng.integers(0, n, len(rows)).astype(np.uint64)
        m = oracle.build_csr(n, n, rows, cols)
        mr, mc = m.pairs()
        n_dp = int(rng.choice([0, 1, 30, 2000]))
        n_dm = int(rng.choice([0, 1, 30, 2000]))
        pr = rng.integers(0, n, n_dp).astype(np.uint64)
        pc = rng.integers(0, n, n_dp).astype(np.uint64)
        kill = rng.choice(m.nnz, min(n_dm, m.nnz), replace=False) if m.nnz else np.array([], dtype=np.int64)
        dr = np.concatenate([mr[kill], pr[: n_dp // 3]])      # some tombstones also shadow pending adds
        dc = np.concatenate([mc[kill], pc[: n_dp // 3]])
        dp = oracle.build_csr(n, n, pr, pc) if n_dp else None
        dm = oracle.build_csr(n, n, dr, dc) if len(dr) else None
        M = ctx.mat_from_csr(n, n, m.rowptr, m.colidx)
        DP = ctx.mat_from_coo(n, n, pr, pc) if n_dp else None
        DM = ctx.mat_from_coo(n, n, dr, dc) if len(dr) else None
        for masks_dp in (False, True):
            got = M.merge(DP, DM, dm_masks_dp=masks_dp)
            ref = oracle.merge(m, dp, dm, masks_dp)
            rp, ci, _ = got.export_csr()
            assert got.nvals == ref.nnz, (case, masks_dp)
            np.testing.assert_array_equal(rp, ref.rowptr, err_msg=f"case {case} masks_dp {masks_dp}")
            np.testing.assert_array_equal(ci, ref.colidx, err_msg=f"case {case} masks_dp {masks_dp}")


def test_result_arrays_keep_their_context_alive_until_they_are_dropped():
    """Result arrays are zero-copy views of pinned blocks the context owns (Context._take): closing the context while one
    is alive must not leave it dangling — the finalize waits for the last view."""
    import gc
    c = engine.Context(0)
    rng = np.random.default_rng(5)
    r, cc = rand_coo(rng, 300, 300, 4000)
    a = oracle.build_csr(300, 300, r, cc)
    m = c.mat_from_coo(300, 300, r, cc)
    rp, ci, _ = m.export_csr()
    m.free()
    c.close()                                    # put off: rp / ci are alive
    assert c.handle and c._close_pending and c._live_views >= 2
    np.testing.assert_array_equal(rp, a.rowptr)  # still the library's memory, still valid
    np.testing.assert_array_equal(ci, a.colidx)
    del rp, ci
    gc.collect()
    assert not c.handle and c._live_views == 0   # the last view finalised the context
