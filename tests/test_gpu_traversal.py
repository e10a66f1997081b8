"""GPU parity: ANY_PAIR products, CondTraverse expansion core and BFS vs the CPU oracle.
Bit-exact (integer/structural work): identical CSR arrays, identical level vectors; parents
are checked for validity because LAGraph's ANY monoid leaves the choice open."""
import numpy as np
import pytest

import oracle
from falkordb_amd import engine

pytestmark = pytest.mark.gpu
U64 = np.uint64


def up(ctx, a: oracle.CSR):
    return ctx.mat_from_csr(a.nrows, a.ncols, a.rowptr, a.colidx)


def assert_same(mat, ref: oracle.CSR):
    rp, ci, _ = mat.export_csr()
    assert mat.nvals == ref.nnz
    np.testing.assert_array_equal(rp, ref.rowptr)
    np.testing.assert_array_equal(ci, ref.colidx)


def f_matrix(k, n, srcs):
    rows = np.arange(k, dtype=U64)
    return oracle.build_csr(k, n, rows, np.asarray(srcs, dtype=U64))


@pytest.mark.parametrize("scale,k", [(10, 64), (14, 1024), (16, 5000)])
def test_lmxm_single_and_multi_source(ctx, scale, k):
    # Matrix::lmxm (matrix.rs:930-947): hop 1 (one source per row) then hop 2 (multi-source rows)
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rng = np.random.default_rng(scale)
    f = f_matrix(k, n, rng.integers(0, n, k))
    A, F = up(ctx, a), up(ctx, f)
    c1_ref, _ = oracle.mxm(f, a)
    C1 = F.mxm(A)
    assert_same(C1, c1_ref)
    c2_ref, _ = oracle.mxm(c1_ref, a)
    assert_same(C1.mxm(A), c2_ref)


def test_lmxm_hub_rows_hit_bitmap_class(ctx):
    # a product row far above 4096 gathered entries exercises the global-bitmap sort class
    a = oracle.rmat_csr(15)
    deg = np.diff(a.rowptr).astype(np.int64)
    hubs = np.argsort(-deg)[:8]
    f = oracle.build_csr(2, a.nrows, np.repeat(np.arange(2, dtype=U64), 4), hubs.astype(U64))
    c1, _ = oracle.mxm(f, a)
    c2, _ = oracle.mxm(c1, a)
    A = up(ctx, a)
    assert_same(up(ctx, f).mxm(A).mxm(A), c2)


def _delta_layers(a: oracle.CSR, rng, n_dm, n_dp):
    rows, cols = a.pairs()
    pick = rng.choice(len(rows), n_dm, replace=False)
    dm = oracle.build_csr(a.nrows, a.ncols, rows[pick], cols[pick])
    pr = rng.integers(0, a.nrows, n_dp, dtype=np.uint64)
    pc = rng.integers(0, a.ncols, n_dp, dtype=np.uint64)
    dp = oracle.build_csr(a.nrows, a.ncols, pr, pc)
    return dp, dm


@pytest.mark.parametrize("n_dm,n_dp", [(0, 50), (50, 0), (300, 300)])
def test_delta_lmxm_all_branches(ctx, n_dm, n_dp):
    # Matrix::delta_lmxm (matrix.rs:1317-1402) incl. the row-level mask quirk on hop 2
    a = oracle.rmat_csr(12)
    rng = np.random.default_rng(n_dm * 7 + n_dp)
    dp, dm = _delta_layers(a, rng, n_dm, n_dp)
    k = 700
    f = f_matrix(k, a.nrows, rng.integers(0, a.nrows, k))
    A, DP, DM, F = up(ctx, a), up(ctx, dp), up(ctx, dm), up(ctx, f)
    h1_ref, _ = oracle.delta_lmxm(f, a, dp, dm)
    H1 = F.delta_lmxm(A, DP, DM)
    assert_same(H1, h1_ref)
    h2_ref, _ = oracle.delta_lmxm(h1_ref, a, dp, dm)
    assert_same(H1.delta_lmxm(A, DP, DM), h2_ref)


def test_expand_chain_with_label_filter_and_skipped_rows(ctx):
    # expand_batch core (cond_traverse.rs:452-751): skipped sources leave empty rows, 3 hops,
    # dst-label bitmap post-filter, ascending (row, dest)
    a = oracle.rmat_csr(11)
    n = a.nrows
    rng = np.random.default_rng(42)
    dp, dm = _delta_layers(a, rng, 100, 100)
    k = 300
    src = rng.integers(0, n, k).astype(U64)
    src[::7] = np.uint64(2**64 - 1)
    valid = src != np.uint64(2**64 - 1)
    f = oracle.build_csr(k, n, np.arange(k, dtype=U64)[valid], src[valid])
    label_ids = np.nonzero((oracle.mix64(np.arange(n, dtype=U64)) % np.uint64(3)) == 0)[0]
    label = oracle.bits_from_ids(n, label_ids)
    A, DP, DM = up(ctx, a), up(ctx, dp), up(ctx, dm)
    c, flops_ref = f, 0
    for _ in range(3):
        c, fl = oracle.delta_lmxm(c, a, dp, dm)
        flops_ref += fl
    rows, cols = c.pairs()
    keep = np.isin(cols, label_ids)
    ref = oracle.build_csr(k, n, rows[keep], cols[keep])
    rp, dest, flops = engine.expand(ctx, src, [A, A, A], [DP, DP, DP], [DM, DM, DM], label)
    np.testing.assert_array_equal(rp, ref.rowptr)
    np.testing.assert_array_equal(dest, ref.colidx)
    assert flops == flops_ref
    nnz, cs, fl2 = engine.expand_count(ctx, src, [A, A, A], [DP, DP, DP], [DM, DM, DM], label)
    assert nnz == ref.nnz and cs == oracle.checksum(ref) and fl2 == flops_ref
    # fgpu_expand_mat: the same F left on the device as a matrix handle (cond_traverse.rs:602-608) — whole export, a row
    # window through fgpu_mat_extract (what F.iter(min_row, max_row) walks, :644), point probes, in every expand_mode
    for mode in (0, 1, 2):
        try:
            ctx.set_option("expand_mode", mode)
            Fm, fl3 = engine.expand_mat(ctx, src, [A, A, A], [DP, DP, DP], [DM, DM, DM], label)
        finally:
            ctx.set_option("expand_mode", 0)
        assert fl3 == flops_ref and (Fm.nrows, Fm.ncols, Fm.nvals) == (k, n, ref.nnz)
        frp, fci, _ = Fm.export_csr()
        np.testing.assert_array_equal(frp, ref.rowptr)
        np.testing.assert_array_equal(fci, ref.colidx)
        wr, wc, _ = Fm.extract(40, 99)
        lo, hi = int(ref.rowptr[40]), int(ref.rowptr[100])
        np.testing.assert_array_equal(wc, ref.colidx[lo:hi])
        np.testing.assert_array_equal(wr, np.repeat(np.arange(40, 100, dtype=U64), np.diff(ref.rowptr[40:101]).astype(np.int64)))
        Fm.free()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 64, 65, 300, 1100, 4200])
def test_expand_modes_match_oracle(ctx, mode, k):
    """fgpu_expand through both product forms (sorted CSR / bit-parallel, bitexpand.hip) and the per-hop
    automatic choice: identical (rowptr, dest), flops and checksum, with delta layers (row-level mask
    quirk included), skipped source rows, repeated sources and a destination-label filter.  k spans one
    word, a word boundary, a padded row stride (1100 -> 18 words -> 32) and a stride beyond one
    wavefront (4200 -> 66 words -> 128)."""
    a = oracle.rmat_csr(10)
    n = a.nrows
    rng = np.random.default_rng(1000 * mode + k)
    dp, dm = _delta_layers(a, rng, 60, 60)
    src = rng.integers(0, n, k).astype(U64)   # repeats on purpose
    if k > 8:
        src[::5] = np.uint64(2**64 - 1)
    valid = src != np.uint64(2**64 - 1)
    f = oracle.build_csr(k, n, np.arange(k, dtype=U64)[valid], src[valid])
    label_ids = np.nonzero((oracle.mix64(np.arange(n, dtype=U64)) % np.uint64(4)) != 0)[0]
    label = oracle.bits_from_ids(n, label_ids)
    A, DP, DM = up(ctx, a), up(ctx, dp), up(ctx, dm)
    ctx.set_option("expand_mode", mode)
    try:
        for hops, with_delta, with_label in [(1, True, False), (2, False, True), (3, True, True)]:
            c, flops_ref = f, 0
            for _ in range(hops):
                c, fl = oracle.delta_lmxm(c, a, dp if with_delta else None, dm if with_delta else None)
                flops_ref += fl
            if with_label:
                rows, cols = c.pairs()
                keep = np.isin(cols, label_ids)
                c = oracle.build_csr(k, n, rows[keep], cols[keep])
            mats = [A] * hops
            dps = [DP] * hops if with_delta else None
            dms = [DM] * hops if with_delta else None
            rp, dest, flops = engine.expand(ctx, src, mats, dps, dms, label if with_label else None)
            np.testing.assert_array_equal(rp, c.rowptr)
            np.testing.assert_array_equal(dest, c.colidx)
            assert flops == flops_ref
            nnz, cs, fl = engine.expand_count(ctx, src, mats, dps, dms, label if with_label else None)
            assert nnz == c.nnz and cs == oracle.checksum(c) and fl == flops_ref
            # the counting last hop (rows counted where they are produced) against the separate count pass, and the
            # count-only form
            ctx.set_option("expand_fuse_count", 0)
            assert engine.expand_count(ctx, src, mats, dps, dms, label if with_label else None) == (nnz, cs, fl)
            ctx.set_option("expand_fuse_count", 1)
            assert engine.expand_count(ctx, src, mats, dps, dms, label if with_label else None,
                                       want_checksum=False) == (nnz, 0, fl)
    finally:
        ctx.set_option("expand_mode", 0)
        ctx.set_option("expand_fuse_count", 1)


@pytest.mark.parametrize("k", [1, 3, 100, 200, 500, 1000])
@pytest.mark.parametrize("dirty", [False, True])
def test_sparse_pull_row_groups_match_the_item_form_and_the_oracle(ctx, k, dirty):
    """The sparse mid-chain pull of the bit-parallel form (bp_pull_groups_kernel: a wavefront per 32 rows of A', split
    rows left to the item kernel) on a graph built to hit its corners: a vertex count that is not a multiple of 32 or
    64, in-degrees of 0, 1, exactly 256 (one item), 257 (the smallest split row), ~600 and > 4096 (hub chunks), rows
    inside one group that mix all of these; k spans every row stride 1 / 2 / 4 / 8 / 16 words.  expand_mode 2 puts the
    first hop in bit form, where X holds only the sources: the sparse pull runs.  Both forms, the oracle, flops."""
    n = 5000 + 17
    rng = np.random.default_rng(77 + k)
    rows, cols = [], []
    def fan_in(dst, deg):
        srcs = rng.choice(n, deg, replace=False)
        rows.extend(srcs.tolist()); cols.extend([dst] * deg)
    for dst, deg in [(0, 256), (1, 257), (2, 600), (31, 1), (32, 4500), (33, 255), (63, 300), (n - 1, 258), (n - 2, 256),
                     (4990, 5000), (100, 64), (101, 65), (102, 63)]:
        fan_in(dst, deg)
    extra = 20000
    rows.extend(rng.integers(0, n, extra).tolist()); cols.extend(rng.integers(0, n, extra).tolist())
    a = oracle.build_csr(n, n, np.array(rows, dtype=U64), np.array(cols, dtype=U64))
    dp, dm = _delta_layers(a, rng, 80, 80) if dirty else (None, None)
    src = rng.integers(0, n, k).astype(U64)
    f = oracle.build_csr(k, n, np.arange(k, dtype=U64), src)
    A = up(ctx, a)
    DP, DM = (up(ctx, dp), up(ctx, dm)) if dirty else (None, None)
    ctx.set_option("expand_mode", 2)
    try:
        for hops in (1, 2, 3):
            c, flops_ref = f, 0
            for _ in range(hops):
                c, fl = oracle.delta_lmxm(c, a, dp, dm)
                flops_ref += fl
            mats = [A] * hops
            dps = [DP] * hops if dirty else None
            dms = [DM] * hops if dirty else None
            got = {}
            for groups in (1, 0):
                ctx.set_option("expand_row_groups", groups)
                rp, dest, flops = engine.expand(ctx, src, mats, dps, dms)
                np.testing.assert_array_equal(rp, c.rowptr)
                np.testing.assert_array_equal(dest, c.colidx)
                assert flops == flops_ref
                got[groups] = engine.expand_count(ctx, src, mats, dps, dms)
                assert got[groups] == (c.nnz, oracle.checksum(c), flops_ref)
            assert got[0] == got[1]
    finally:
        ctx.set_option("expand_mode", 0)
        ctx.set_option("expand_row_groups", 1)


@pytest.mark.parametrize("layers", ["both", "dp", "dm"])
@pytest.mark.parametrize("k", [1, 7, 700, 4000])
def test_dirty_first_hop_from_one_entry_rows_matches_the_general_delta_product(ctx, layers, k):
    """The first hop of a batch over DIRTY layers (first_hop_rows_dirty: (m[u] \\ dm[u]) U dp[u] per row, candidates placed
    by rank) against the oracle's delta_lmxm (matrix.rs:1323-1361 restated) and against the general three-product path
    (expand_first_hop = 0), on deltas that also BREAK the Delta invariants the engine must not rely on: pending additions m
    already holds, tombstones of entries m never had, an entry named by both layers; hub rows, repeated sources."""
    a = oracle.rmat_csr(12)
    n = a.nrows
    rng = np.random.default_rng(500 + k)
    rows, cols = a.pairs()
    pick = rng.choice(len(rows), 3000, replace=False)
    dm_r = np.concatenate([rows[pick], rng.integers(0, n, 500, dtype=np.uint64)])          # + tombstones outside m
    dm_c = np.concatenate([cols[pick], rng.integers(0, n, 500, dtype=np.uint64)])
    dup = rng.choice(len(rows), 400, replace=False)
    dp_r = np.concatenate([rng.integers(0, n, 3000, dtype=np.uint64), rows[dup], rows[pick[:300]]])   # + held by m, + in dm too
    dp_c = np.concatenate([rng.integers(0, n, 3000, dtype=np.uint64), cols[dup], cols[pick[:300]]])
    dm = oracle.build_csr(n, n, dm_r, dm_c) if layers != "dp" else None
    dp = oracle.build_csr(n, n, dp_r, dp_c) if layers != "dm" else None
    deg = np.diff(a.rowptr).astype(np.int64)
    hubs = np.argsort(-deg)[:4].astype(U64)
    src = rng.integers(0, n, k).astype(U64)
    src[: min(k, 4)] = hubs[: min(k, 4)]
    if k > 10:
        src[5] = src[6]                                                                      # the same source in two rows
    f = oracle.build_csr(k, n, np.arange(k, dtype=U64), src)
    A = up(ctx, a)
    DP = up(ctx, dp) if dp is not None else None
    DM = up(ctx, dm) if dm is not None else None
    try:
        for hops in (1, 2, 3):
            c, flops_ref = f, 0
            for _ in range(hops):
                c, fl = oracle.delta_lmxm(c, a, dp, dm)
                flops_ref += fl
            got = {}
            for fh in (1, 0):
                ctx.set_option("expand_first_hop", fh)
                rp, dest, flops = engine.expand(ctx, src, [A] * hops, [DP] * hops, [DM] * hops)
                np.testing.assert_array_equal(rp, c.rowptr)
                np.testing.assert_array_equal(dest, c.colidx)
                assert flops == flops_ref
                got[fh] = engine.expand_count(ctx, src, [A] * hops, [DP] * hops, [DM] * hops)
                assert got[fh] == (c.nnz, oracle.checksum(c), flops_ref)
            assert got[0] == got[1]
    finally:
        ctx.set_option("expand_first_hop", 1)


@pytest.mark.parametrize("k,with_delta,with_label", [(1, False, False), (70, True, False), (300, True, True),
                                                     (1100, False, True)])
def test_expand_levels_per_hop_sets_and_distinct_union(ctx, k, with_delta, with_label):
    """fgpu_expand_levels (variable-length [*1..4] core, SURVEY §8f-1 / BASELINE config 5): per-hop result of the
    delta_lmxm chain and the DISTINCT union over the hops, against the oracle's chain of products."""
    a = oracle.rmat_csr(10)
    n = a.nrows
    rng = np.random.default_rng(77 + k)
    dp, dm = _delta_layers(a, rng, 50, 50)
    src = rng.integers(0, n, k).astype(U64)
    if k > 8:
        src[::7] = np.uint64(2**64 - 1)
    valid = src != np.uint64(2**64 - 1)
    f = oracle.build_csr(k, n, np.arange(k, dtype=U64)[valid], src[valid])
    label_ids = np.nonzero((oracle.mix64(np.arange(n, dtype=U64)) % np.uint64(3)) != 0)[0]
    label = oracle.bits_from_ids(n, label_ids)
    A, DP, DM = up(ctx, a), up(ctx, dp), up(ctx, dm)
    hops = 4
    got = engine.expand_levels(ctx, src, [A] * hops, [DP] * hops if with_delta else None,
                               [DM] * hops if with_delta else None, label if with_label else None)
    c, union, flops_ref = f, set(), 0
    for h in range(hops):
        c, fl = oracle.delta_lmxm(c, a, dp if with_delta else None, dm if with_delta else None)
        flops_ref += fl
        rows, cols = c.pairs()
        if with_label:
            keep = np.isin(cols, label_ids)
            rows, cols = rows[keep], cols[keep]
        shown = oracle.build_csr(k, n, rows, cols)
        assert got["hop_nnz"][h] == shown.nnz, f"hop {h + 1}"
        assert got["hop_checksum"][h] == oracle.checksum(shown), f"hop {h + 1}"
        union |= set(zip(rows.tolist(), cols.tolist()))
    assert got["flops"] == flops_ref
    u = oracle.build_csr(k, n, [r for r, _ in union], [c_ for _, c_ in union])
    assert got["union_nnz"] == len(union) and got["union_checksum"] == oracle.checksum(u)
    if not with_delta and not with_label:
        # clean layers: the DISTINCT [*1..4] end points of one source = BFS levels 1..4 (+ the source itself
        # when it lies on a cycle of length <= 4)
        for i in np.nonzero(valid)[0][:5]:
            level, _, _ = oracle.bfs(a, int(src[i]), 4, want_parent=False)
            reach = {int(v) for v in np.nonzero((level >= 1) & (level <= 4))[0]}
            mine = {c_ for r, c_ in union if r == i}
            assert mine - {int(src[i])} == reach - {int(src[i])}


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("dirty", [False, True])
def test_trail_counts_for_one_and_two_hops_match_the_dfs(ctx, weighted, dirty):
    """fgpu_expand_trail_counts (exact path counts for k <= 2) against the oracle's DFS over trails
    (cond_var_len_traverse.rs:196-387): a small cyclic graph with self-loops, reciprocal edges, a hub, and — weighted —
    multi-edges given as per-pair multiplicities; clean and with pending adds / tombstones."""
    from oracle import model
    rng = np.random.default_rng(5 + weighted + 2 * dirty)
    n = 40
    pairs = {(int(a), int(b)) for a, b in rng.integers(0, n, (260, 2))}
    pairs |= {(3, 3), (7, 7), (3, 7), (7, 3), (0, 0)} | {(5, int(b)) for b in range(0, n, 2)}
    mult = {p: (int(rng.integers(1, 4)) if weighted else 1) for p in pairs}
    base = sorted(pairs)
    dm_pairs = [base[i] for i in rng.choice(len(base), 25, replace=False)] if dirty else []
    dp_pairs = [(int(a), int(b)) for a, b in rng.integers(0, n, (30, 2)) if (int(a), int(b)) not in pairs] if dirty else []
    dp_mult = {p: (int(rng.integers(1, 4)) if weighted else 1) for p in set(dp_pairs)}

    def mat(ps, w):
        ps = sorted(set(ps))
        r = np.array([p[0] for p in ps], dtype=U64)
        c = np.array([p[1] for p in ps], dtype=U64)
        v = np.array([w[p] for p in ps], dtype=U64) if weighted else None
        return ctx.mat_from_coo(n, n, r, c, v)

    M = mat(base, mult)
    DP = mat(dp_pairs, dp_mult) if dp_pairs else None
    DM = ctx.mat_from_coo(n, n, np.array([p[0] for p in dm_pairs], dtype=U64), np.array([p[1] for p in dm_pairs], dtype=U64)) if dm_pairs else None
    eff = {p: mult[p] for p in pairs if p not in set(dm_pairs)}
    eff.update(dp_mult)
    edges, eid = [], 0
    for (a, b), w in sorted(eff.items()):
        for _ in range(w):
            edges.append((eid, a, b))
            eid += 1
    src = np.array([3, 7, 0, 5, 11, 39, 3], dtype=U64)
    for k in (1, 2):
        rowptr, dest, count = engine.expand_trail_counts(ctx, src, [M] * k, [DP] * k if dirty else None,
                                                         [DM] * k if dirty else None, weighted=weighted)
        for i, s_ in enumerate(src.tolist()):
            want = model.trail_counts(edges, s_, k)
            got = dict(zip(dest[rowptr[i]:rowptr[i + 1]].tolist(), count[rowptr[i]:rowptr[i + 1]].tolist()))
            assert got == want, (k, s_)
            assert list(dest[rowptr[i]:rowptr[i + 1]]) == sorted(want)       # ascending destinations
    with pytest.raises(Exception):
        engine.expand_trail_counts(ctx, src, [M] * 3)                       # no product form beyond two hops
    if DP is not None:
        with pytest.raises(Exception):
            engine.expand_trail_counts(ctx, src, [M, DP])    # two different layers: not ONE var-length relationship


def test_trail_counts_on_rmat_agree_with_the_products(ctx):
    """On a graph without self-loops the 2-hop trail count of (row, dest) is the number of intermediates: the sum of the
    counts is the traversed-edge count of the second hop and the support is the 2-hop reachable set."""
    a = oracle.rmat_csr(12)
    A = up(ctx, a)
    src = np.random.default_rng(8).choice(a.nrows, 300, replace=False).astype(U64)
    rowptr, dest, count = engine.expand_trail_counts(ctx, src, [A, A])
    f = f_matrix(len(src), a.nrows, src)
    c1, _ = oracle.mxm(f, a)
    c2, fl2 = oracle.mxm(c1, a)
    assert np.array_equal(rowptr, c2.rowptr) and np.array_equal(dest, c2.colidx)
    assert int(count.sum()) == fl2 and int(count.min()) >= 1


def test_expand_rmat22_three_hops_both_forms_agree(ctx):
    """Full-size graph (BASELINE.json configs[1]/[2] shape): 512 sources, 3 hops — the sorted-CSR chain
    and the bit-parallel chain must give the same result size, order-independent checksum and flops."""
    A = ctx.mat_rmat(22)
    n = A.nrows
    src = np.random.default_rng(3).choice(n, 512, replace=False).astype(U64)
    out = {}
    try:
        for mode in (1, 2, 0):
            ctx.set_option("expand_mode", mode)
            out[mode] = engine.expand_count(ctx, src, [A, A, A])
    finally:
        ctx.set_option("expand_mode", 0)
    assert out[1] == out[2] == out[0]
    assert out[1][0] > 10_000_000


def check_bfs(a: oracle.CSR, level, parent, src, ref_level):
    np.testing.assert_array_equal(level, ref_level)
    if parent is None:
        return
    assert parent[src] == src
    reached = np.nonzero(level > 0)[0]
    p = parent[reached]
    assert np.all(p >= 0)
    assert np.all(level[p] + 1 == level[reached])          # parent sits one level up
    s = a.to_set() if a.nnz < 2_000_000 else None
    if s is not None:
        assert all((int(pp), int(v)) in s for pp, v in zip(p, reached))  # and the edge exists
    assert np.all(parent[level < 0] == -1)


@pytest.mark.parametrize("force", [0, 1, 2])
@pytest.mark.parametrize("scale", [8, 13, 16])
def test_bfs_levels_and_parents(ctx, scale, force):
    a = oracle.rmat_csr(scale)
    A = up(ctx, a)
    At = A.transpose()
    deg = np.diff(a.rowptr)
    roots = np.nonzero(deg > 0)[0][:3]
    plan = engine.BfsPlan(ctx, A, At)
    plan.tune(force_direction=force)
    for src in roots:
        ref_level, _, ref_edges = oracle.bfs(a, int(src), -1)
        plan.run(int(src), -1, want_parent=True)
        level, parent = plan.fetch(want_parent=True)
        check_bfs(a, level, parent, int(src), ref_level)
        st = plan.stats()
        assert st["edges_traversed"] == ref_edges
        assert st["reached"] == int(np.count_nonzero(ref_level >= 0))


@pytest.mark.parametrize("force", [0, 1])
@pytest.mark.parametrize("scale,min_edges", [(9, 1), (13, 64), (16, 2000), (18, 20000)])
def test_bfs_heavy_push_levels_by_propagation_blocking(ctx, scale, min_edges, force):
    """Direction 3 of the fused level loop (bfs.hip bfs_pb_*: the frontier's edges binned by destination window, a workgroup
    per window marks its discoveries in LDS): forced on for small graphs (bfs_pb = 2) with a low edge threshold so that
    the push levels with that many edges go that way (queue-listed frontiers and, through bfs_pb_list_kernel, bitmap ones); levels, parents, traversed edges and reached
    count against the oracle's BFS, with and without parents, auto and push-only direction; max_level truncation; and the
    option off gives the same vectors.  `bfs_pb_last_levels` says the path ran."""
    a = oracle.rmat_csr(scale)
    A = up(ctx, a)
    At = A.transpose()
    deg = np.diff(a.rowptr)
    roots = [int(np.nonzero(deg > 0)[0][0]), int(np.argmax(deg)), int(np.nonzero(deg > 0)[0][-1])]
    try:
        ctx.set_option("bfs_pb", 2)
        ctx.set_option("bfs_pb_min_edges", min_edges)
        plan = engine.BfsPlan(ctx, A, At)
        plan.tune(force_direction=force)
        ran = 0
        for src in roots:
            ref_level, _, ref_edges = oracle.bfs(a, src, -1)
            for want_parent in (True, False):
                plan.run(src, -1, want_parent=want_parent)
                level, parent = plan.fetch(want_parent=want_parent)
                check_bfs(a, level, parent, src, ref_level)
                st = plan.stats()
                assert st["edges_traversed"] == ref_edges
                assert st["reached"] == int(np.count_nonzero(ref_level >= 0))
                ran += ctx.get_option("bfs_pb_last_levels")
            plan.run(src, 2)
            np.testing.assert_array_equal(plan.fetch()[0], np.where((ref_level >= 0) & (ref_level <= 2), ref_level, -1))
            plan.run_async(src, -1, False, 3)                # too few levels enqueued: the top-up launches carry no bfs_pb_* kernels
            plan.wait()
            np.testing.assert_array_equal(plan.fetch()[0], ref_level)
        # (with the direction picked on the device a small graph may pull every level the launches are armed for)
        assert ran > 0 or force == 0, "no level went by propagation blocking"
        plan.free()
        ctx.set_option("bfs_pb", 0)
        plan = engine.BfsPlan(ctx, A, At)
        plan.run(roots[1], -1, want_parent=True)
        level, parent = plan.fetch(want_parent=True)
        check_bfs(a, level, parent, roots[1], oracle.bfs(a, roots[1], -1)[0])
        plan.stats()
        assert ctx.get_option("bfs_pb_last_levels") == 0
        plan.free()
    finally:
        ctx.set_option("bfs_pb", 1)
        ctx.set_option("bfs_pb_min_edges", 2 << 20)


def test_bfs_listed_frontiers_and_listed_candidate_pulls(ctx):
    """The two other jobs of bfs_pb_list_kernel (plans with the propagation-blocking launches; forced on here): a push level
    whose frontier is a sparse bitmap is listed into the queue first, and a pull level with few rows left to discover runs as
    a pull of listed candidates (direction 4, bfs_lp_kernel) — every search against the oracle's levels, parents valid, and
    `bfs_cp_last_mask` says that some of the searches took those paths."""
    took = 0
    try:
        ctx.set_option("bfs_pb", 2)
        ctx.set_option("bfs_pb_min_edges", 1 << 40)            # (no blocked push here: the other two paths alone)
        for scale, ef in ((16, 16), (18, 8), (19, 4)):
            a = oracle.rmat_csr(scale, ef)
            A = up(ctx, a)
            At = A.transpose()
            deg = np.diff(a.rowptr)
            roots = np.nonzero(deg > 0)[0]
            roots = [int(roots[i]) for i in (0, len(roots) // 3, len(roots) // 2, len(roots) - 1)] + [int(np.argmax(deg))]
            plan = engine.BfsPlan(ctx, A, At)
            for rep in range(2):                                # (the second round runs with the launches the first one learnt)
                for src in roots:
                    ref_level, _, ref_edges = oracle.bfs(a, src, -1)
                    want_parent = (src + rep) % 2 == 0
                    plan.run(src, -1, want_parent=want_parent)
                    level, parent = plan.fetch(want_parent=want_parent)
                    check_bfs(a, level, parent, src, ref_level)
                    st = plan.stats()
                    assert st["edges_traversed"] == ref_edges
                    assert st["reached"] == int(np.count_nonzero(ref_level >= 0))
                    took += 1 if ctx.get_option("bfs_cp_last_mask") else 0
            plan.free()
            At.free()
            A.free()
        assert took > 0, "no search listed a frontier or a candidate set"
    finally:
        ctx.set_option("bfs_pb", 1)
        ctx.set_option("bfs_pb_min_edges", 2 << 20)


@pytest.mark.parametrize("max_level", [0, 1, 2, 3])
def test_bfs_max_level(ctx, max_level):
    a = oracle.rmat_csr(12)
    A = up(ctx, a)
    src = int(np.argmax(np.diff(a.rowptr)))
    ref_level, _, _ = oracle.bfs(a, src, max_level)
    level, parent, _ = engine.bfs(ctx, A, A.transpose(), src, max_level, want_parent=True)
    check_bfs(a, level, parent, src, ref_level)


def test_one_call_bfs_keeps_and_replaces_its_plan(ctx):
    """fgpu_bfs (the reference's one-function entry) keeps the plan of the last (A, At) pair on A: repeated calls, a call
    with parents after one without, another transpose handle, no transpose, an option change, a second adjacency over the
    same transpose and the release of the transpose before the adjacency all answer like the oracle."""
    a = oracle.rmat_csr(13)
    A = up(ctx, a)
    At = A.transpose()
    deg = np.diff(a.rowptr)
    roots = [int(x) for x in np.nonzero(deg > 0)[0][:4]]

    def check(A_, At_, src, want_parent, max_level=-1):
        ref_level, _, ref_edges = oracle.bfs(a, src, max_level)
        level, parent, edges = engine.bfs(ctx, A_, At_, src, max_level, want_parent=want_parent)
        if want_parent:
            check_bfs(a, level, parent, src, ref_level)
        else:
            np.testing.assert_array_equal(level, ref_level)
        if max_level < 0:
            assert edges == ref_edges

    for i, src in enumerate(roots):                 # same pair: the plan is reused, with and without parents
        check(A, At, src, want_parent=bool(i & 1))
    check(A, At, roots[0], True, max_level=2)
    At2 = A.transpose()                             # another handle of the transpose: the plan is rebuilt over it
    check(A, At2, roots[1], True)
    check(A, None, roots[2], False)                 # push only
    check(A, At, roots[3], True)
    ctx.set_option("bfs_hub_first", 0)              # options are read at plan creation: the cached plan must not survive
    check(A, At, roots[0], True)
    ctx.set_option("bfs_hub_first", 1)
    check(A, At, roots[1], False)
    B = up(ctx, a)                                  # a second adjacency over the same transpose takes the link over
    check(B, At, roots[2], True)
    check(A, At, roots[3], True)
    At.free()                                       # the transpose goes first: A's (or B's) plan is dropped with it
    check(A, At2, roots[0], True)
    check(B, None, roots[1], False)
    At2.free()
    check(A, None, roots[2], False)


def test_bfs_push_only_without_transpose(ctx):
    a = oracle.rmat_csr(12)
    A = up(ctx, a)
    src = int(np.argmax(np.diff(a.rowptr)))
    ref_level, _, ref_edges = oracle.bfs(a, src, -1)
    level, parent, edges = engine.bfs(ctx, A, None, src, -1, want_parent=False)
    np.testing.assert_array_equal(level, ref_level)
    assert edges == ref_edges


def test_bfs_chain_graph_many_levels(ctx):
    # a path 0->1->...->299: 299 levels, exercises the blind level batching and termination
    n = 300
    a = oracle.build_csr(n, n, np.arange(n - 1, dtype=U64), np.arange(1, n, dtype=U64))
    A = up(ctx, a)
    level, _, _ = engine.bfs(ctx, A, A.transpose(), 0, -1, want_parent=False)
    np.testing.assert_array_equal(level, np.arange(n, dtype=np.int32))


def test_bfs_async_pipeline_two_plans(ctx):
    """run_async / wait: two plans alternate so search i+1 is enqueued while search i runs; `levels`
    smaller than the search depth exercises the enqueue-more path of wait()."""
    a = oracle.rmat_csr(14)
    A = up(ctx, a)
    At = A.transpose()
    roots = [int(r) for r in np.nonzero(np.diff(a.rowptr) > 0)[0][:6]]
    plans = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
    for levels in (0, 2, 12):
        refs = {}
        for i, src in enumerate(roots):
            plans[i % 2].run_async(src, -1, False, levels)
            if i > 0:
                prev = roots[i - 1]
                plans[(i - 1) % 2].wait()
                lv, _ = plans[(i - 1) % 2].fetch()
                refs.setdefault(prev, oracle.bfs(a, prev, -1)[0])
                np.testing.assert_array_equal(lv, refs[prev])
        plans[(len(roots) - 1) % 2].wait()
        lv, _ = plans[(len(roots) - 1) % 2].fetch()
        np.testing.assert_array_equal(lv, oracle.bfs(a, roots[-1], -1)[0])


def test_bfs_hub_source_and_isolated_source(ctx):
    # the queue / hub-census path: a source that is itself a hub row (>= 4096 out-edges), and one
    # with no out-edges at all (the search ends after level 0)
    n = 6000
    rows = np.concatenate([np.zeros(5000, dtype=U64), np.arange(1, 5001, dtype=U64)])
    cols = np.concatenate([np.arange(1, 5001, dtype=U64), (np.arange(1, 5001, dtype=U64) % 900) + 5001])
    a = oracle.build_csr(n, n, rows, cols)
    A = up(ctx, a)
    At = A.transpose()
    for src in (0, 5999, 17):
        ref, _, ref_edges = oracle.bfs(a, src, -1)
        level, parent, edges = engine.bfs(ctx, A, At, src, -1, want_parent=True)
        check_bfs(a, level, parent, src, ref)
        assert edges == ref_edges


def test_vxm_auto_direction_picks_push_or_the_tiled_pull(ctx):
    """direction 0: a sparse frontier is pushed, a dense one goes through the LDS-tiled pull (the kernel of the RMAT-22
    boolean-SpMV roofline case) — same bits as the oracle either way, with and without a mask."""
    a = oracle.rmat_csr(15)
    n = a.nrows
    A = up(ctx, a)
    At = A.transpose()
    rng = np.random.default_rng(40)
    mask = oracle.bits_from_ids(n, rng.choice(n, n // 4, replace=False))
    with pytest.raises(Exception):
        At.tiles_info()                                       # no tiles yet
    for k in (7, n // 64):                                    # both below 1 / 32 of the vertices: pushed, no tiles built
        f = oracle.bits_from_ids(n, rng.choice(n, k, replace=False))
        np.testing.assert_array_equal(engine.vxm(ctx, f, None, A, At, 0), oracle.vxm(a, f, None))
    with pytest.raises(Exception):
        At.tiles_info()
    for k in (n // 16, n // 2, n):                            # dense: the tiled pull
        f = oracle.bits_from_ids(n, rng.choice(n, k, replace=False))
        np.testing.assert_array_equal(engine.vxm(ctx, f, mask, A, At, 0), oracle.vxm(a, f, mask))
        np.testing.assert_array_equal(engine.vxm(ctx, f, None, A, None, 0), oracle.vxm(a, f, None))   # no At: pushed
    assert At.tiles_info()["items"] > 0                       # the dense calls built the layout on At


@pytest.mark.parametrize("direction", [1, 2])
def test_vxm_masked(ctx, direction):
    a = oracle.rmat_csr(13)
    n = a.nrows
    rng = np.random.default_rng(direction)
    f = oracle.bits_from_ids(n, rng.choice(n, 500, replace=False))
    mask = oracle.bits_from_ids(n, rng.choice(n, n // 3, replace=False))
    A = up(ctx, a)
    At = A.transpose()
    for mk in (None, mask):
        want = oracle.vxm(a, f, mk)
        got = engine.vxm(ctx, f, mk, A, At, direction)
        np.testing.assert_array_equal(got, want)


def test_bfs_rmat22_properties(ctx):
    """BASELINE.json configs[1] at full size, through size-independent properties (the serial oracle takes
    seconds per root here): level(src) = 0; along every edge level[v] <= level[u] + 1 for reached u;
    every reached v != src has an in-neighbour one level up (checked through the returned parents);
    reached count and traversed-edge count agree with the level vector; the async two-plan path agrees
    with the synchronous one."""
    A = ctx.mat_rmat(22)
    At = A.transpose()
    rp, ci, _ = A.export_csr()
    rp = rp.astype(np.int64)
    ci = ci.astype(np.int64)
    deg = np.diff(rp)
    n = A.nrows
    src_of_edge = np.repeat(np.arange(n, dtype=np.int64), deg)
    plan = engine.BfsPlan(ctx, A, At)
    plan2 = engine.BfsPlan(ctx, A, At)
    for src in [int(np.nonzero(deg > 0)[0][0]), int(np.argmax(deg))]:
        plan.run(src, -1, want_parent=True)
        level, parent = plan.fetch(want_parent=True)
        st = plan.stats()
        reached = level >= 0
        assert level[src] == 0 and parent[src] == src
        assert st["reached"] == int(reached.sum())
        assert st["edges_traversed"] == int(deg[reached].sum())
        lu, lv = level[src_of_edge], level[ci]
        live = lu >= 0
        assert (lv[live] >= 0).all() and (lv[live] <= lu[live] + 1).all()
        others = reached.copy()
        others[src] = False
        p = parent[others]
        assert (level[p] + 1 == level[others]).all()
        # the parent edge exists: binary search parent's row for the child
        child = np.nonzero(others)[0]
        lo, hi = rp[p], rp[p + 1]
        pos = np.array([np.searchsorted(ci[a:b], c) for a, b, c in zip(lo[:2000], hi[:2000], child[:2000])])
        assert (ci[lo[:2000] + pos] == child[:2000]).all()
        assert (parent[~reached] == -1).all()
        plan2.run_async(src, -1, False, 4)
        plan2.wait()
        lv2, _ = plan2.fetch()
        np.testing.assert_array_equal(lv2, level)


def test_bfs_rmat22_directions_agree_over_repeated_runs(ctx):
    """Race regression: in a pull level the hub section of a workgroup that ran ahead sets a hub row's visited
    and next-frontier bits; the wave that owns the row's 64-bit word must then OR its own discoveries in, not
    store them (a plain store dropped the hub from the next frontier and ~50 of its children were never
    reached, a few runs in a hundred).  Push-only levels are the reference (atomics only); auto and pull-only
    searches must reproduce them on every repetition, and max_level searches must be their truncation."""
    A = ctx.mat_rmat(22)
    At = A.transpose()
    rp, _, _ = A.export_csr()
    deg = np.diff(rp.astype(np.int64))
    roots = np.nonzero(deg > 0)[0][:10].tolist()
    ref_plan = engine.BfsPlan(ctx, A, At)
    ref_plan.tune(force_direction=1)
    plans = {0: engine.BfsPlan(ctx, A, At), 2: engine.BfsPlan(ctx, A, At)}
    plans[2].tune(force_direction=2)
    for src in roots:
        ref_plan.run(src)
        ref = ref_plan.fetch()[0].copy()
        nlev = ref_plan.stats()["levels"]
        for force, plan in plans.items():
            for rep in range(4 if force == 0 else 2):
                plan.run(src)
                np.testing.assert_array_equal(plan.fetch()[0], ref, err_msg=f"root {src} force {force} rep {rep}")
                st = plan.stats()
                assert st["reached"] == int((ref >= 0).sum())
                assert st["edges_traversed"] == int(deg[ref >= 0].sum())
        for k in (nlev - 2, nlev - 1):
            plans[0].run(src, k)
            np.testing.assert_array_equal(plans[0].fetch()[0], np.where((ref >= 0) & (ref <= k), ref, -1))


@pytest.mark.parametrize("tiny", [0, 1, 2])
def test_bfs_tiny_kernel_modes(ctx, tiny):
    """bfs_tiny_kernel (consecutive tiny levels in one single-workgroup launch; option bfs_tiny: off / on / when the
    previous search was deep): a 3000-level path, and R-MAT searches whose first and last levels are tiny, every
    search run three times so that the history-sized blind sequences are exercised; levels, parents, max_level."""
    ctx.set_option("bfs_tiny", tiny)
    try:
        n = 3000
        path = oracle.build_csr(n, n, np.arange(n - 1, dtype=U64), np.arange(1, n, dtype=U64))
        P = up(ctx, path)
        plan = engine.BfsPlan(ctx, P, P.transpose())
        for rep in range(3):
            plan.run(0, -1, want_parent=True)
            level, parent = plan.fetch(want_parent=True)
            np.testing.assert_array_equal(level, np.arange(n, dtype=np.int32))
            np.testing.assert_array_equal(parent[1:], np.arange(n - 1))
            st = plan.stats()
            assert st["levels"] == n and st["reached"] == n and st["edges_traversed"] == n - 1
        plan.run(0, 1234)
        level, _ = plan.fetch()
        np.testing.assert_array_equal(level, np.where(np.arange(n) <= 1234, np.arange(n), -1))
        a = oracle.rmat_csr(15)
        A = ctx.mat_rmat(15)
        plan = engine.BfsPlan(ctx, A, A.transpose())
        deg = np.diff(a.rowptr)
        for src in [int(np.argmax(deg)), int(np.nonzero(deg == 1)[0][0]), 7]:
            for max_level in (-1, 1, 3):
                ref, _, _ = oracle.bfs(a, src, max_level)
                for rep in range(3):
                    plan.run(src, max_level, want_parent=True)
                    level, parent = plan.fetch(want_parent=True)
                    np.testing.assert_array_equal(level, ref)
                    check_bfs(a, level, parent, src, ref)
    finally:
        ctx.set_option("bfs_tiny", 2)


# ---- the result path of the two ABI entries the reference would call (ctx.hip: pinned pool, DMA, streamed chunks) ----------

@pytest.fixture(scope="module")
def rmat18(ctx):
    a = oracle.rmat_csr(18)
    return up(ctx, a), a


@pytest.mark.parametrize("pinned", [1, 0])
def test_expand_host_arrays_pinned_and_staged_paths_agree_with_the_oracle(ctx, rmat18, pinned):
    """fgpu_expand hands back (rowptr, dest) in host memory: pinned pool blocks filled by one DMA each, ids widened on the
    device (pinned_results = 1), or the caller's allocator + the staging ring (0, the round-3 path).  Both against the
    oracle's chain, a result large enough (> 256 KiB) for the pinned route to be the one taken."""
    A, a = rmat18
    src = np.arange(11, a.nrows, 997, dtype=U64)
    c, flops, _ = oracle.expand_omp(src, [(a, None, None)] * 2)
    assert c.nnz * 8 > (1 << 20)
    try:
        ctx.set_option("pinned_results", pinned)
        for _ in range(3):                                   # the second and third results reuse the first one's blocks
            rowptr, dest, fl = engine.expand(ctx, src, [A, A])
            assert fl == flops
            np.testing.assert_array_equal(rowptr, c.rowptr)
            np.testing.assert_array_equal(dest, c.colidx)
            del rowptr, dest
    finally:
        ctx.set_option("pinned_results", 1)


@pytest.mark.parametrize("emit_sort", [1, 2, 0])
def test_expand32_and_both_emission_forms_agree_with_the_oracle(ctx, rmat18, emit_sort):
    """fgpu_expand32 — the result in the device's own 32-bit form, two DMAs, nothing widened — entry for entry against the
    oracle's chain, with the bit state turned into rows by pairs + the stable sort (expand_emit_sort = 2, round 6), by the
    ballot transpose of rounds 3-5 (0) and by whichever the density of the result picks (1, the default: the labelled 2-hop
    result here is sparse, the others hold > 8 entries per vertex and take the ballot transpose after the pairs' count pass);
    2 and 3 hops (CSR and bit-form chains), dirty layers, NULL rows and a label."""
    A, a = rmat18
    rng = np.random.default_rng(7)
    dm = oracle.sample(a, 0x18D, 500)
    DM = up(ctx, dm)
    src = np.arange(11, a.nrows, 499, dtype=U64)
    src[5] = np.iinfo(np.uint64).max
    label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(5) != 0
    bits = oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])
    try:
        ctx.set_option("expand_emit_sort", emit_sort)
        for hops, layers, dev in ((2, [(a, None, None)] * 2, ([A, A],)), (3, [(a, None, dm)] * 3, ([A] * 3, None, [DM] * 3))):
            keep = src != np.iinfo(np.uint64).max
            f = oracle.build_csr(len(src), a.nrows, np.nonzero(keep)[0].astype(U64), src[keep])
            fl = 0
            for (m_, dp_, dm_) in layers:
                f, x = oracle.delta_lmxm_omp(f, m_, dp_, dm_, 8)
                fl += x
            rowptr, dest, flops = engine.expand32(ctx, src, *dev)
            assert rowptr.dtype == np.uint32 and dest.dtype == np.uint32 and flops == fl
            np.testing.assert_array_equal(rowptr, f.rowptr.astype(np.uint32))
            np.testing.assert_array_equal(dest, f.colidx.astype(np.uint32))
            assert f.nnz > 1_000_000
            assert f.nnz > 8 * a.nrows                            # (the dense side of the density rule, bitexpand.hip BP_DENSE_OUT)
            if hops == 2:
                rowptr, dest, _ = engine.expand32(ctx, src, *dev, dst_label_bitmap=bits)
                sel = label[f.colidx.astype(np.int64)]
                assert 1_000_000 < int(sel.sum()) < 8 * a.nrows       # (... and its sparse side: pairs + sort)
                np.testing.assert_array_equal(dest, f.colidx[sel].astype(np.uint32))
                rows = np.repeat(np.arange(f.nrows), np.diff(f.rowptr).astype(np.int64))[sel]
                np.testing.assert_array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=f.nrows))]).astype(np.uint32))
    finally:
        ctx.set_option("expand_emit_sort", 1)
        DM.free()


@pytest.mark.parametrize("chunk_rows,dest_bits", [(1, 64), (7, 32), (64, 64), (4096, 32)])
def test_expand_stream_chunks_concatenate_to_the_oracle_result(ctx, rmat18, chunk_rows, dest_bits):
    """fgpu_expand_stream_*: what CondTraverseOp::expand_batch walks (cond_traverse.rs:644-751), chunk by chunk — the
    chunks cover the source rows in order, their relative row pointers and destinations concatenate to the oracle's
    (rowptr, dest), for chunk sizes below, at and above the batch size and both id widths; empty rows included."""
    A, a = rmat18
    src = np.arange(5, a.nrows, 1531, dtype=U64)
    c, flops, _ = oracle.expand_omp(src, [(a, None, None)] * 2)
    st = engine.ExpandStream(ctx, src, [A, A], chunk_rows=chunk_rows, dest_bits=dest_bits)
    assert st.nnz == c.nnz and st.flops == flops
    nxt, dest, rowptr = 0, [], [np.zeros(1, dtype=U64)]
    for first, rp, d in st:
        assert first == nxt and 1 <= len(rp) - 1 <= chunk_rows and rp[0] == 0
        assert d.dtype == (np.uint64 if dest_bits == 64 else np.uint32)
        rowptr.append(rp[1:].astype(U64) + rowptr[-1][-1])
        dest.append(d.astype(U64))                            # copied: the views die with the next step
        nxt = first + len(rp) - 1
    st.close()
    assert nxt == len(src)
    np.testing.assert_array_equal(np.concatenate(rowptr), c.rowptr)
    np.testing.assert_array_equal(np.concatenate(dest) if dest else np.zeros(0, dtype=U64), c.colidx)
    # closing early, and a batch with no result at all
    st = engine.ExpandStream(ctx, src, [A, A], chunk_rows=3)
    next(st)
    st.close()
    e = ctx.mat_new(a.nrows, a.nrows)
    st = engine.ExpandStream(ctx, src[:5], [e])
    got = list(st)
    assert st.nnz == 0 and sum(len(rp) - 1 for _, rp, _ in got) == 5 and all(len(d) == 0 for _, _, d in got)
    st.close()


@pytest.mark.parametrize("row_bits", [16, 32])
def test_expand_pairs_columns_match_the_oracle_with_and_without_pinned_rows(ctx, rmat18, row_bits):
    """fgpu_expand_pairs: the (active_row, dest) columns of CondTraverseOp::expand_batch (cond_traverse.rs:644-751) built on
    the device — every pair of the oracle's chain in (row, dest) order; with pre-bound `to` values (:657-661) a pinned row
    keeps exactly its pinned destination when the chain reaches it and nothing otherwise, free rows are untouched; a
    destination label filters first; a batch with no result and a pinned id no vertex carries."""
    A, a = rmat18
    src = np.arange(5, a.nrows, 1531, dtype=U64)
    k = len(src)
    c, flops, _ = oracle.expand_omp(src, [(a, None, None)] * 2)
    ref_rows = np.repeat(np.arange(k, dtype=U64), np.diff(c.rowptr).astype(np.int64))
    rows, dest, fl = engine.expand_pairs(ctx, src, [A, A], row_bits=row_bits)
    assert fl == flops
    np.testing.assert_array_equal(rows, ref_rows)
    np.testing.assert_array_equal(dest, c.colidx)
    # pinned rows: every third row pinned to one of its own destinations, every fifth to a vertex it does not reach
    rng = np.random.default_rng(7)
    pin = np.full(k, 2 ** 64 - 1, dtype=U64)
    want_r, want_d = [], []
    for i in range(k):
        row = c.colidx[int(c.rowptr[i]):int(c.rowptr[i + 1])]
        if i % 3 == 0 and len(row):
            pin[i] = row[rng.integers(0, len(row))]
            want_r.append(i); want_d.append(int(pin[i]))
        elif i % 5 == 0:
            miss = int(rng.integers(0, a.nrows))
            while miss in set(row.tolist()):
                miss = int(rng.integers(0, a.nrows))
            pin[i] = miss
        elif i % 7 == 0:
            pin[i] = 2 ** 40 + 3                             # no vertex carries this id
        else:
            want_r += [i] * len(row); want_d += row.tolist()
    rows, dest, fl = engine.expand_pairs(ctx, src, [A, A], pinned_dest=pin, row_bits=row_bits)
    assert fl == flops                                       # the chain ran whole: the pinning filters its result
    np.testing.assert_array_equal(rows, np.asarray(want_r, dtype=U64))
    np.testing.assert_array_equal(dest, np.asarray(want_d, dtype=U64))
    # a destination label (applied by the chain) and pinned rows together
    label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(2) == 0
    bits = oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])
    rows, dest, _ = engine.expand_pairs(ctx, src, [A, A], dst_label_bitmap=bits, pinned_dest=pin, row_bits=row_bits)
    keep = label[np.asarray(want_d, dtype=np.int64)]
    np.testing.assert_array_equal(rows, np.asarray(want_r, dtype=U64)[keep])
    np.testing.assert_array_equal(dest, np.asarray(want_d, dtype=U64)[keep])
    e = ctx.mat_new(a.nrows, a.nrows)
    rows, dest, _ = engine.expand_pairs(ctx, src[:5], [e], row_bits=row_bits)
    assert len(rows) == 0 and len(dest) == 0


def test_bfs_into_pinned_caller_arrays_matches_the_oracle(ctx, rmat18):
    """fgpu_bfs with level[] / parent[] in pinned memory from fgpu_host_alloc (DMA, the parent widened on the device)
    against the same call into pageable numpy arrays (staging ring) and the oracle's levels."""
    A, a = rmat18
    At = A.transpose()
    n = a.nrows
    level_p, parent_p = ctx.host_array(n, np.int32), ctx.host_array(n, np.int64)
    for src in (int(np.argmax(np.diff(a.rowptr))), 3):
        ref_level, _, ref_edges = oracle.bfs(a, src, -1)
        level_p[:] = 77
        parent_p[:] = 77
        lv, par, e = engine.bfs(ctx, A, At, src, -1, True, level_out=level_p, parent_out=parent_p)
        assert lv is level_p and e == ref_edges
        np.testing.assert_array_equal(level_p, ref_level)
        lv2, par2, _ = engine.bfs(ctx, A, At, src, -1, True)
        np.testing.assert_array_equal(lv2, ref_level)
        reached = ref_level > 0
        for par_ in (parent_p, par2):                          # any valid parent: one level up, -1 if unreached
            assert (ref_level[par_[reached]] + 1 == ref_level[reached]).all() and (par_[ref_level < 0] == -1).all()
            assert par_[src] == src
    del level_p, parent_p
    # the pool hands the block back
    again = ctx.host_array(n, np.int32)
    again[:] = 1
    assert int(again.sum()) == n
