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


@pytest.fixture(scope="module")
def rmat20_refs(rmat20):
    """(nnz, checksum, flops) of the 3-hop chain at RMAT-20 for source sets of 100 / 160 / 400 / 640 rows — bit rows of 2 /
    4 (three in use) / 8 / 16 words, every width the partitioned count hop serves — clean and dirty; for 400 rows also under
    a destination label."""
    A, dp, dm, a, hdp, hdm = rmat20
    allsrc = p_sources(a.nrows, 1024)
    label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(3) != 0
    refs = {}
    for nsrc in (100, 160, 400, 640):
        src = allsrc[:nsrc]
        for dirty in (False, True):
            c, flops, _ = oracle.expand_omp(src, [(a, hdp, hdm) if dirty else (a, None, None)] * 3)
            refs[(nsrc, dirty, False)] = (c.nnz, oracle.checksum_omp(c), flops)
            if nsrc == 400:
                keep = label[c.colidx.astype(np.int64)]
                rows = np.repeat(np.arange(c.nrows), np.diff(c.rowptr).astype(np.int64))[keep]
                rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=c.nrows))]).astype(U64)
                cl = oracle.CSR(c.nrows, c.ncols, rp, c.colidx[keep])
                refs[(nsrc, dirty, True)] = (cl.nnz, oracle.checksum_omp(cl), flops)
            del c
    return allsrc, refs, oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])


@pytest.mark.parametrize("xcd,nsrc", [(1, 100), (1, 160), (1, 400), (1, 640), (0, 400)])
def test_khop_rmat20_count_hop_partitioned_by_xcd(ctx, rmat20, rmat20_refs, xcd, nsrc):
    """The dense count hop in its XCD-partitioned form (bitpart.hip: every entry of A' gathered by the XCD that owns its row of
    X, partial rows folded per vertex) against the oracle's (nnz, checksum, flops): bit rows of 2 / 4 / 8 / 16 words, clean and
    dirty layers (the delta fix-ups act on the folded rows of the side buffer), count-only and checksum forms, a destination
    label; xcd = 0 is the plain pull on the same inputs (the A/B the bench quotes).  expand_xcd_min_mb = 0 forces the form on a
    state that would otherwise be too small to bother."""
    A, dp, dm, a, hdp, hdm = rmat20
    allsrc, refs, label_bits = rmat20_refs
    src = allsrc[:nsrc]
    try:
        ctx.set_option("expand_mode", 2)
        ctx.set_option("expand_xcd", xcd)
        ctx.set_option("expand_xcd_min_mb", 0)
        for dirty in (False, True):
            layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
            for _ in range(2):                                 # the second call reuses the plan (and the recycled state)
                assert engine.expand_count(ctx, src, *layers) == refs[(nsrc, dirty, False)], (xcd, nsrc, dirty)
            nn, _, fl = engine.expand_count(ctx, src, *layers, want_checksum=False)
            assert (nn, fl) == (refs[(nsrc, dirty, False)][0], refs[(nsrc, dirty, False)][2])
            if nsrc == 400:
                lay = layers if dirty else ([A] * 3, None, None)
                assert engine.expand_count(ctx, src, *lay, dst_label_bitmap=label_bits) == refs[(nsrc, dirty, True)], (xcd, dirty, "label")
    finally:
        ctx.set_option("expand_mode", 0)
        ctx.set_option("expand_xcd", 1)
        ctx.set_option("expand_xcd_min_mb", 32)


@pytest.fixture(scope="module")
def probe_cases(rmat20):
    """Per (hops, dirty): 200 :P sources, a bound destination for each — drawn from the row itself, from the vertices the
    tombstones removed, at random, beyond the matrix — and whether the oracle's delta_lmxm chain reaches it."""
    A, dp, dm, a, hdp, hdm = rmat20
    k = 200
    src = p_sources(a.nrows, k, first=7000)
    cases = {}
    for hops in (1, 2, 3):
        clean_c = oracle.expand_omp(src, [(a, None, None)] * hops)[0]
        for dirty in (False, True):
            c = oracle.expand_omp(src, [(a, hdp, hdm)] * hops)[0] if dirty else clean_c
            rng = np.random.default_rng(100 * hops + dirty)
            dst = np.zeros(k, dtype=U64)
            want = np.zeros(k, dtype=bool)
            for i in range(k):
                row = c.colidx[int(c.rowptr[i]):int(c.rowptr[i + 1])]
                crow = clean_c.colidx[int(clean_c.rowptr[i]):int(clean_c.rowptr[i + 1])]
                r = i % 4
                if r == 0 and len(row):
                    dst[i] = row[rng.integers(0, len(row))]                 # reached
                elif r == 1 and len(crow):
                    dst[i] = crow[rng.integers(0, len(crow))]               # reached over clean layers: maybe tombstoned now
                elif r == 2:
                    dst[i] = rng.integers(0, a.nrows)
                else:
                    dst[i] = a.nrows + 5 if i % 8 == 3 else rng.integers(0, a.nrows)
                j = np.searchsorted(row, dst[i])
                want[i] = j < len(row) and row[j] == dst[i]
            assert want.sum() > k // 5 and (~want).sum() > k // 5
            cases[(hops, dirty)] = (dst, want)
    return src, cases


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("hops", [1, 2, 3])
def test_expand_probe_all_rows_pinned_matches_the_oracle(ctx, rmat20, probe_cases, mode, hops):
    """fgpu_expand_probe — every row of the batch has a pre-bound destination (CondTraverse with `to` bound, the multi-hop
    ExpandInto shape of tests/flow/test_expand_into.py:63-95; cond_traverse.rs:657-661): present[i] must say whether dst[i] is
    in row i of the oracle's delta_lmxm chain — for 1 / 2 / 3 hops, clean and dirty layers (the row-level tombstone mask of
    matrix.rs:1343-1361 decides single entries here), with the chain kept in sorted-CSR form (mode 1: the last hop is a binary
    search per frontier entry), in bit form from the first hop (mode 2: one bit of one row of the state) and left to choose;
    under a destination label; with an unbound source."""
    A, dp, dm, a, hdp, hdm = rmat20
    src, cases = probe_cases
    k = len(src)
    label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(4) != 0
    bits = oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])
    for dirty in (False, True):
        dst, want = cases[(hops, dirty)]
        gm = [A] * hops
        gl = ([dp] * hops, [dm] * hops) if dirty else (None, None)
        try:
            ctx.set_option("expand_mode", mode)
            got, _ = engine.expand_probe(ctx, src, dst, gm, *gl)
            np.testing.assert_array_equal(got, want)
            got, _ = engine.expand_probe(ctx, src, dst, gm, *gl, dst_label_bitmap=bits)
            inb = dst < a.nrows
            lw = want.copy()
            lw[inb] &= label[dst[inb].astype(np.int64)]
            np.testing.assert_array_equal(got, lw)
            s2 = src.copy()
            s2[5] = 2 ** 64 - 1                                          # an unbound source: its row is empty
            got, _ = engine.expand_probe(ctx, s2, dst, gm, *gl)
            w2 = want.copy(); w2[5] = False
            np.testing.assert_array_equal(got, w2)
        finally:
            ctx.set_option("expand_mode", 0)


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


def test_khop_rmat24_clean_matches_the_oracle(ctx, bench_graphs):
    """BASELINE config 3 at full size (RMAT-24, 268 M edges, 3 hops, clean layers): 64 sources of the :P set —
    nnz, checksum and flops of fgpu_expand_count against the oracle, plus the per-hop sizes through
    fgpu_expand_levels."""
    A, _, a = bench_graphs(24)
    src = p_sources(a.nrows, 64)
    c, flops, hop_nnz = oracle.expand_omp(src, [(a, None, None)] * 3)
    ref = (c.nnz, oracle.checksum_omp(c), flops)
    del c
    got = engine.expand_count(ctx, src, [A] * 3)
    assert got == ref, (got, ref)
    lv = engine.expand_levels(ctx, src, [A] * 3)
    assert list(lv["hop_nnz"]) == hop_nnz and lv["flops"] == flops
    assert ref[0] > 50_000_000


def hypersparse(ctx, m):
    """The layer as Delta<T> keeps it (versioned_matrix.rs:214-235): a row list + a short row-pointer array."""
    rp, ci, _ = m.export_csr()
    deg = np.diff(rp.astype(np.int64))
    rows = np.nonzero(deg)[0].astype(U64)
    short = np.concatenate([[0], np.cumsum(deg[deg > 0])]).astype(U64)
    return ctx.mat_from_csr(m.nrows, m.ncols, short, ci, hyper_rows=rows)


def bench_batch(ctx, bench_graphs, scale, rows):
    """Exactly what bench.py times at this scale (bench.py khop_inputs / khop_headline): the session's RMAT-<scale> graph,
    dm = fgpu_mat_sample(0xD3170 + scale, 1000), dp = seeded random coordinates outside A, both hypersparse; the first
    `rows` :P sources of batch 0.  The oracle chains (clean, dirty) are computed once, 64 source rows at a time."""
    A, _, a = bench_graphs(scale)
    dp, dm, hdp, hdm = bench_graphs.khop_layers(scale)
    src = p_sources(A.nrows, 1024)[:rows]
    ref_clean = oracle.expand_summary_omp(src, [(a, None, None)] * 3)
    ref_dirty = oracle.expand_summary_omp(src, [(a, hdp, hdm)] * 3)
    return A, dp, dm, src, ref_clean, ref_dirty, (a, hdp, hdm)


@pytest.fixture(scope="module")
def rmat24_bench(ctx, bench_graphs):
    return bench_batch(ctx, bench_graphs, 24, 1024)


@pytest.fixture(scope="module")
def rmat22_bench(ctx, bench_graphs):
    return bench_batch(ctx, bench_graphs, 22, 1024)


@pytest.fixture(scope="module")
def rmat26_bench(ctx, bench_graphs):
    return bench_batch(ctx, bench_graphs, 26, 128)


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("dirty", [False, True])
def test_khop_rmat22_headline_batch_matches_the_oracle(ctx, rmat22_bench, mode, dirty):
    """The bench line's own workload (BASELINE metric: k-hop MATCH at RMAT scale 22): timed batch 0 = the first 1024 :P
    sources, 3 hops, clean and dirty layers, expand_mode auto and bit-parallel — (nnz, checksum, flops) of
    fgpu_expand_count, the count-only form and the per-hop sizes against the oracle's delta_lmxm chain
    (matrix.rs:1317-1402 driven as cond_traverse.rs:600-651 drives it)."""
    A, dp, dm, src, ref_clean, ref_dirty, _ = rmat22_bench
    ref = ref_dirty if dirty else ref_clean
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
    try:
        ctx.set_option("expand_mode", mode)
        got = engine.expand_count(ctx, src, *layers)
        nn, _, fl = engine.expand_count(ctx, src, *layers, want_checksum=False)
        lv = engine.expand_levels(ctx, src, *layers)
    finally:
        ctx.set_option("expand_mode", 0)
    assert got == ref[:3], (mode, dirty, got, ref)
    assert (nn, fl) == (ref[0], ref[2])
    assert list(lv["hop_nnz"]) == ref[3] and lv["flops"] == ref[2]
    assert ref[0] > 300_000_000


def test_khop_rmat22_2048_source_batch_matches_the_oracle(ctx, bench_graphs):
    """bench.py's secondary figure `b2048`: 2048 :P sources per call (about 1000 of them live: the compacted bit state fills
    the 128-byte rows) — (nnz, checksum, flops) and the per-hop sizes against the oracle's chain over the same 2048 rows,
    and 3000 sources (rows wider than one 128-byte block) likewise."""
    A22, _, a22 = bench_graphs(22)
    a18 = oracle.rmat_csr(18)
    A18 = ctx.mat_from_csr(a18.nrows, a18.ncols, a18.rowptr, a18.colidx)
    for A, a, count in ((A22, a22, 2048), (A18, a18, 3000)):    # (the oracle's chain over 3000 rows of RMAT-22 is half a minute)
        src = p_sources(A.nrows, count)
        ref = oracle.expand_summary_omp(src, [(a, None, None)] * 3)
        got = engine.expand_count(ctx, src, [A] * 3)
        lv = engine.expand_levels(ctx, src, [A] * 3)
        assert got == ref[:3], (count, got, ref[:3])
        assert list(lv["hop_nnz"]) == ref[3] and lv["flops"] == ref[2]


@pytest.mark.parametrize("dirty", [False, True])
def test_khop_rmat26_batch_rows_match_the_oracle(ctx, rmat26_bench, dirty):
    """The metric's other scale (RMAT-26, 1.06 G edges): the first 128 rows of batch 0 (the oracle's chain for 1024 rows
    is a minute of CPU at this size), 3 hops, clean and dirty, auto and bit-parallel modes."""
    A, dp, dm, src, ref_clean, ref_dirty, _ = rmat26_bench
    ref = ref_dirty if dirty else ref_clean
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
    for mode in (0, 2):
        try:
            ctx.set_option("expand_mode", mode)
            got = engine.expand_count(ctx, src, *layers)
            lv = engine.expand_levels(ctx, src, *layers)
        finally:
            ctx.set_option("expand_mode", 0)
        assert got == ref[:3], (mode, dirty, got, ref)
        assert list(lv["hop_nnz"]) == ref[3] and lv["flops"] == ref[2]
    assert ref[0] > 100_000_000


@pytest.mark.parametrize("dirty", [False, True])
def test_khop_rmat26_full_batch_matches_the_committed_oracle_run(ctx, bench_graphs, dirty):
    """RMAT-26 at the size it is quoted: ALL 1024 rows of batch 0, 3 hops, clean and dirty — (nnz, checksum, flops) and the
    per-hop sizes against tests/golden/khop26_batch0.json, the CPU oracle's chain over exactly these inputs run once by
    tests/golden/make_khop26_golden.py (45 + 103 s of CPU: not repeated per session).  The inputs are re-derived here and
    pinned by hashes, so a change of the generator or of the layers cannot pass silently."""
    import hashlib
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "khop26_batch0.json")))
    A, _, a = bench_graphs(26)
    dp, dm, hdp, hdm = bench_graphs.khop_layers(26)
    src = p_sources(A.nrows, 1024)
    assert gold["rows"] == 1024 and gold["edges"] == a.nnz and gold["nnz_dp"] == hdp.nnz and gold["nnz_dm"] == hdm.nnz
    assert gold["sources_sha256"] == hashlib.sha256(np.ascontiguousarray(src).tobytes()).hexdigest()
    if not dirty:                                            # (4.3 GB: hashed once)
        assert gold["colidx_sha256"] == hashlib.sha256(np.ascontiguousarray(a.colidx).tobytes()).hexdigest()
    ref = gold["dirty" if dirty else "clean"]
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
    got = engine.expand_count(ctx, src, *layers)
    assert got == (ref["nnz"], ref["checksum"], ref["flops"]), (dirty, got, ref)
    nn, _, fl = engine.expand_count(ctx, src, *layers, want_checksum=False)
    assert (nn, fl) == (ref["nnz"], ref["flops"])
    lv = engine.expand_levels(ctx, src, *layers)
    assert list(lv["hop_nnz"]) == ref["hop_nnz"] and lv["flops"] == ref["flops"]
    assert ref["nnz"] > 4_000_000_000


def _scan_opts(ctx, **kw):
    """set whole-frontier options, return the old values (restored by the caller's finally)"""
    old = {k: ctx.get_option(k) for k in kw}
    for k, v in kw.items():
        ctx.set_option(k, v)
    return old


@pytest.mark.parametrize("dirty", [False, True])
def test_whole_frontier_call_matches_the_oracle(ctx, rmat20, dirty):
    """fgpu_expand_count over a whole frontier (spgemm.hip expand_count_scan; SURVEY.md §7 hard part 1, §8d: all sources of a
    scan in one call): 6000 source rows of RMAT-20 — NULL rows, a repeated source and sources without out-edges among them —
    3 hops, clean and dirty, with and without a destination label: (nnz, checksum, flops) against the oracle's delta_lmxm
    chain over the SAME rows (every row hashed by its index in the call), for every pass width / lane count the option
    allows here, in the auto and the bit-parallel mode, count-only included."""
    A, dp, dm, a, hdp, hdm = rmat20
    src = p_sources(a.nrows, 6000).copy()
    src[17] = np.iinfo(np.uint64).max                      # a NULL source row (cond_traverse.rs:566-575, OPTIONAL)
    src[4000] = np.iinfo(np.uint64).max
    src[123] = src[122]                                    # the same node bound on two rows
    lay = [(a, hdp, hdm) if dirty else (a, None, None)] * 3
    keep = np.nonzero(src != np.iinfo(np.uint64).max)[0]
    # the oracle's F holds one row per call row: NULL rows stay empty (their index still counts)
    ref = None
    nnz = cs = flops = 0
    for c0 in range(0, len(keep), 64):
        rows = keep[c0:c0 + 64]
        f = oracle.build_csr(len(src), a.nrows, rows.astype(U64), src[rows])
        for (m_, dp_, dm_) in lay:
            f, fl = oracle.delta_lmxm_omp(f, m_, dp_, dm_, 8)
            flops += fl
        nnz += f.nnz
        cs = (cs + oracle.checksum_omp(f, 8)) & 0xFFFFFFFFFFFFFFFF
    ref = (nnz, cs, flops)
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
    seen = set()
    for rows_, lanes, mode in ((1024, 3, 0), (1024, 1, 0), (256, 4, 0), (512, 2, 2), (64, 3, 0)):
        old = _scan_opts(ctx, expand_scan_rows=rows_, expand_scan_lanes=lanes, expand_mode=mode)
        try:
            got = engine.expand_count(ctx, src, *layers)
            nn, _, fl = engine.expand_count(ctx, src, *layers, want_checksum=False)
            passes, live = ctx.get_option("expand_scan_last_passes"), ctx.get_option("expand_scan_last_live")
        finally:
            _scan_opts(ctx, **old)
        assert got == ref, (rows_, lanes, mode, got, ref)
        assert (nn, fl) == (ref[0], ref[2])
        assert passes == -(-live // rows_) and 2000 < live < len(keep)
        seen.add(passes)
    assert len(seen) >= 3 and ref[0] > 500_000_000
    # ... and against the calls it replaces: 1024 rows per call (nnz and flops add up; the checksum hashes call rows)
    bn = bf = 0
    for j in range(0, len(src), 1024):
        r = engine.expand_count(ctx, src[j:j + 1024], *layers)
        bn += r[0]
        bf += r[2]
    assert (bn, bf) == (ref[0], ref[2])
    # a destination label (the dst filter of the LAST hop, cond_traverse.rs:647-651)
    if not dirty:
        label = oracle.mix64(np.arange(a.nrows, dtype=U64)) % U64(3) != 0
        bits = oracle.bits_from_ids(a.nrows, np.nonzero(label)[0])
        small = src[:2500]
        got = engine.expand_count(ctx, small, [A] * 3, dst_label_bitmap=bits)
        want = [0, 0]
        for c0 in range(0, len(small), 1024):
            r = engine.expand_count(ctx, small[c0:c0 + 1024], [A] * 3, dst_label_bitmap=bits)
            want = [want[0] + r[0], want[1] + r[2]]
        assert [got[0], got[2]] == want and 0 < got[0]


def test_whole_frontier_call_on_a_graph_that_stays_in_csr_form(ctx):
    """The same call where the chain never leaves the sorted-CSR products (a small graph: every pass ends in CSR form and is
    check-summed with the call's row indices), and in expand_mode 1 where the whole-frontier path is not taken at all."""
    a = oracle.rmat_csr(12)
    A = ctx.mat_from_csr(a.nrows, a.ncols, a.rowptr, a.colidx)
    src = np.arange(0, a.nrows, dtype=U64)[:3000]
    ref = oracle.expand_summary_omp(src, [(a, None, None)] * 2, chunk=256, threads=8)
    for mode in (0, 1):
        old = _scan_opts(ctx, expand_mode=mode, expand_scan_rows=256)
        try:
            got = engine.expand_count(ctx, src, [A] * 2)
        finally:
            _scan_opts(ctx, **old)
        assert got == ref[:3], (mode, got, ref[:3])
    A.free()


def test_clean_first_hop_over_exactly_4096_rows(ctx, rmat20):
    """a clean first hop over exactly FH_MAX_ROWS = 4096 one-entry rows published an unwritten row pointer (the scan
    kernel's 1024 threads x 4 rows write rp[0..4095]); 4095, 4096 and 4097 rows through the single-call chain (the
    whole-frontier path switched off) against the oracle."""
    A, dp, dm, a, hdp, hdm = rmat20
    allsrc = p_sources(a.nrows, 4097)
    old = _scan_opts(ctx, expand_scan_min=0)
    try:
        for k in (4095, 4096, 4097):
            src = allsrc[:k]
            ref = oracle.expand_summary_omp(src, [(a, None, None)] * 2, chunk=512, threads=8)
            got = engine.expand_count(ctx, src, [A] * 2)
            assert got == ref[:3], (k, got, ref[:3])
    finally:
        _scan_opts(ctx, **old)


def test_khop_rmat22_whole_scan_matches_the_committed_oracle_run(ctx, bench_graphs):
    """The bench line's step at the size it is quoted: ALL :P sources of RMAT-22 (261 6xx rows) in ONE fgpu_expand_count call,
    3 hops, clean — (nnz, checksum, flops) against tests/golden/khop22_scan.json, the CPU oracle's chain over exactly these
    inputs run once by tests/golden/make_khop22_scan_golden.py (a quarter of an hour of 16-thread CPU: not repeated per
    session); then the first 32 slabs of 1024 rows as separate whole-frontier calls against the golden's per-slab sums (nnz,
    flops: the checksum of a slab hashes rows of the WHOLE call).  Inputs re-derived here and pinned by hashes."""
    import hashlib
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "khop22_scan.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/khop22_scan.json not generated yet")
    gold = json.load(open(path))
    A, _, a = bench_graphs(22)
    ids = np.arange(a.nrows, dtype=U64)
    src = ids[oracle.mix64(ids) % U64(16) == 0]
    assert gold["rows"] == len(src) and gold["edges"] == a.nnz
    assert gold["sources_sha256"] == hashlib.sha256(np.ascontiguousarray(src).tobytes()).hexdigest()
    assert gold["colidx_sha256"] == hashlib.sha256(np.ascontiguousarray(a.colidx).tobytes()).hexdigest()
    got = engine.expand_count(ctx, src, [A] * 3)
    assert got == (gold["nnz"], gold["checksum"], gold["flops"]), (got, gold["nnz"], gold["checksum"], gold["flops"])
    assert ctx.get_option("expand_scan_last_passes") > 100
    nn, _, fl = engine.expand_count(ctx, src, [A] * 3, want_checksum=False)
    assert (nn, fl) == (gold["nnz"], gold["flops"])
    S = gold["slab_rows"]
    assert sum(x[0] for x in gold["slabs"]) == gold["nnz"] and sum(x[2] for x in gold["slabs"]) == gold["flops"]
    for j in (0, 1, 7, 31):
        r = engine.expand_count(ctx, src[j * S:(j + 1) * S], [A] * 3)
        assert (r[0], r[2]) == (gold["slabs"][j][0], gold["slabs"][j][2]), j
    r = engine.expand_count(ctx, src[:8 * S], [A] * 3)            # 8 slabs, one call: the prefix sums, checksum included
    assert r == (sum(x[0] for x in gold["slabs"][:8]), sum(x[1] for x in gold["slabs"][:8]) & 0xFFFFFFFFFFFFFFFF,
                 sum(x[2] for x in gold["slabs"][:8]))
    assert gold["nnz"] > 100_000_000_000


def test_khop_rmat24_two_hop_host_arrays_match_the_oracle(ctx, rmat24_bench):
    """What the operator consumes (cond_traverse.rs:608, 644-751): the (row_i, dest) arrays of fgpu_expand in HOST memory
    for the 1024-row 2-hop batch bench.py's materialised leg times (30.8 M entries at RMAT-24), entry for entry against
    the oracle's chain — and the same result as a device matrix (fgpu_expand_mat)."""
    A, dp, dm, src, _, _, (a, hdp, hdm) = rmat24_bench
    c, flops, _ = oracle.expand_omp(src, [(a, None, None)] * 2)
    rowptr, dest, fl = engine.expand(ctx, src, [A] * 2)
    assert fl == flops and len(dest) == c.nnz > 20_000_000
    assert np.array_equal(rowptr, c.rowptr) and np.array_equal(dest, c.colidx)
    m, fl = engine.expand_mat(ctx, src, [A] * 2)
    rp, ci, _ = m.export_csr()
    m.free()
    assert fl == flops and np.array_equal(rp, c.rowptr) and np.array_equal(ci, c.colidx)
    c, flops, _ = oracle.expand_omp(src, [(a, hdp, hdm)] * 2)
    rowptr, dest, fl = engine.expand(ctx, src, [A] * 2, [dp] * 2, [dm] * 2)
    assert fl == flops and np.array_equal(rowptr, c.rowptr) and np.array_equal(dest, c.colidx)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("dirty", [False, True])
def test_khop_rmat24_bench_batch_matches_the_oracle(ctx, rmat24_bench, mode, dirty):
    """The configuration bench.py times, at its size: one 1024-row batch (16 words per vertex row) of the RMAT-24
    3-hop chain, clean and with hypersparse dirty layers on every hop, in each expand_mode — nnz, checksum and
    traversed-edge count of fgpu_expand_count, and the per-hop sizes of fgpu_expand_levels, against the oracle's
    delta_lmxm chain (matrix.rs:1317-1402 driven as cond_traverse.rs:600-605 drives it)."""
    A, dp, dm, src, ref_clean, ref_dirty, (a, hdp, hdm) = rmat24_bench
    ref = ref_dirty if dirty else ref_clean
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if dirty else ([A] * 3,)
    if mode == 1:
        # sorted-CSR products only: the 3-hop product of 1024 rows is 21 G gathered entries — that mode runs the first
        # 64 rows of the batch (the other two modes take the whole batch)
        sub = src[:64]
        want = oracle.expand_summary_omp(sub, [(a, hdp, hdm) if dirty else (a, None, None)] * 3)
        try:
            ctx.set_option("expand_mode", 1)
            got = engine.expand_count(ctx, sub, *layers)
        finally:
            ctx.set_option("expand_mode", 0)
        assert got == want[:3], (got, want)
        return
    try:
        ctx.set_option("expand_mode", mode)
        got = engine.expand_count(ctx, src, *layers)
        nn, _, fl = engine.expand_count(ctx, src, *layers, want_checksum=False)
        lv = engine.expand_levels(ctx, src, *layers)
    finally:
        ctx.set_option("expand_mode", 0)
    assert got == ref[:3], (mode, dirty, got, ref)
    assert (nn, fl) == (ref[0], ref[2])
    assert list(lv["hop_nnz"]) == ref[3] and lv["flops"] == ref[2]
    assert ref[0] > 1_000_000_000          # 1.5 G result entries per batch


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


def test_bfs_rmat26_levels_are_the_bfs_levels(ctx, bench_graphs):
    """BASELINE config 4's graph (RMAT-26, 1.06 G edges) on one device: two roots through the plan API, checked by the
    properties that pin a BFS level vector uniquely — level[src] = 0; no edge (u, v) with u reached skips a level
    (level[v] <= level[u] + 1, v reached); every reached v != src has a parent one level up joined to it by a stored
    edge; reached / edges_traversed agree with the level vector.  The edges are streamed back in row windows so the
    host never holds more than ~3 GB."""
    A, At, _ = bench_graphs(26)
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


def bench_roots(A, want=64):
    """bench.py pick_roots: the first `want` vertex ids with out-degree > 0."""
    rows, _, _ = A.extract(0, 4095)
    return [int(r) for r in np.unique(rows)[:want]]


def test_bfs_rmat22_all_bench_roots_match_the_oracle_levels(ctx, bench_graphs):
    """BASELINE config 2 exactly as bench.py times it: RMAT-22, the 64 roots of the bench, every level vector and
    traversed-edge count against the oracle's BFS (oracle_omp.c orc_bfs_omp, held equal to the serial restatement by
    tests/test_oracle_golden.py) — through the plan API the bench uses (synchronous and the pipelined two-plan loop)
    and through the host-array ABI entry fgpu_bfs."""
    A, At, a = bench_graphs(22)
    trp, tci, _ = At.export_csr()
    at = oracle.CSR(A.nrows, A.ncols, trp, tci)
    roots = bench_roots(A)
    assert len(roots) == 64
    want = {r: oracle.bfs_omp(a, at, r, -1) for r in roots}
    plans = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
    for r in roots:
        plans[0].run(r, -1, False)
        level, _ = plans[0].fetch()
        ref_level, ref_edges = want[r]
        np.testing.assert_array_equal(level, ref_level)
        assert plans[0].stats()["edges_traversed"] == ref_edges
    # the pipelined loop of the timed region: search i + 1 is enqueued while search i runs
    for i, r in enumerate(roots):
        plans[i % 2].run_async(r, -1, False, 0)
        if i > 0:
            plans[(i - 1) % 2].wait()
            level, _ = plans[(i - 1) % 2].fetch()
            np.testing.assert_array_equal(level, want[roots[i - 1]][0])
    plans[(len(roots) - 1) % 2].wait()
    level, _ = plans[(len(roots) - 1) % 2].fetch()
    np.testing.assert_array_equal(level, want[roots[-1]][0])
    for r in roots[:4]:
        level, parent, edges = engine.bfs(ctx, A, At, r, -1, want_parent=True)
        np.testing.assert_array_equal(level, want[r][0])
        assert edges == want[r][1]
        others = level > 0
        assert (level[parent[others]] + 1 == level[others]).all()


def test_bfs_rmat26_two_roots_match_the_oracle_levels(ctx, bench_graphs):
    """BASELINE config 4's graph on one device, level for level against the oracle (push-only OpenMP BFS: the host
    never holds the transposed copy of the 1.06 G-edge graph)."""
    A, At, a = bench_graphs(26)
    plan = engine.BfsPlan(ctx, A, At)
    for r in bench_roots(A, 2):
        ref_level, ref_edges = oracle.bfs_omp(a, None, r, -1)
        plan.run(r, -1, False)
        level, _ = plan.fetch()
        np.testing.assert_array_equal(level, ref_level)
        assert plan.stats()["edges_traversed"] == ref_edges
