"""GPU: the LDS-tiled boolean vxm (tiled.hip) against the CPU oracle and against the CSR pull kernel.

Small tile widths force many column tiles, padding, multi-item hub groups and partially filled
last tiles on graphs the oracle finishes in seconds; the RMAT-22 case (BASELINE.json configs[1])
is checked through the CSR kernel of the same library, itself pinned on the oracle here."""
import numpy as np
import pytest

import oracle
from falkordb_amd import engine

pytestmark = pytest.mark.gpu


def up(ctx, a):
    return ctx.mat_from_csr(a.nrows, a.ncols, a.rowptr, a.colidx)


@pytest.mark.parametrize("tile_bits,vec,k", [(0, 0, 0), (7, 1, 1), (9, 2, 2), (10, 4, 1), (12, 4, 2), (11, 1, 2),
                                             (13, 2, 1), (8, 4, 2)])
def test_tiled_vxm_matches_oracle(ctx, tile_bits, vec, k):
    a = oracle.rmat_csr(13)
    n = a.nrows
    rng = np.random.default_rng(tile_bits * 100 + vec * 10 + k)
    A = up(ctx, a)
    At = A.transpose()
    info = At.build_tiles(tile_bits, vec, k)
    assert info["entries"] >= a.nnz and info["entries"] % info["vec"] == 0
    try:
        for u in (1, 2, 4, 8):
            ctx.set_option("tiled_u", u)
            ctx.set_option("tiled_nt", u == 2)
            for nf in (1, 37, 500, n // 2, n):
                f = oracle.bits_from_ids(n, rng.choice(n, nf, replace=False))
                mask = oracle.bits_from_ids(n, rng.choice(n, n // 3, replace=False))
                for mk in (None, mask):
                    want = oracle.vxm(a, f, mk)
                    for _ in range(3):  # repeated: a timing-dependent hazard once hid behind a single pass
                        got = engine.vxm(ctx, f, mk, A, At, 3)
                        np.testing.assert_array_equal(got, want)
    finally:
        ctx.set_option("tiled_u", 4)
        ctx.set_option("tiled_nt", 0)


@pytest.mark.parametrize("threads", [256, 512, 1024])
def test_tiled_vxm_workgroup_shapes(ctx, threads):
    a = oracle.rmat_csr(12)
    n = a.nrows
    A = up(ctx, a)
    At = A.transpose()
    At.build_tiles(9, 2, 2)
    ctx.set_option("tiled_threads", threads)
    try:
        for wgs in (0, 3, 17):
            ctx.set_option("tiled_wgs", wgs)
            f = oracle.bits_from_ids(n, np.arange(0, n, 3))
            np.testing.assert_array_equal(engine.vxm(ctx, f, None, A, At, 3), oracle.vxm(a, f, None))
    finally:
        ctx.set_option("tiled_threads", 1024)
        ctx.set_option("tiled_wgs", 0)


def test_tiled_vxm_ragged_and_empty(ctx):
    # n not a multiple of 64, empty rows, one hub row longer than any item, an all-empty matrix
    n = 1000
    rows = np.concatenate([np.full(900, 7), np.arange(0, 200, 2), [999]]).astype(np.uint64)
    cols = np.concatenate([np.arange(900) + 50, np.arange(0, 200, 2) + 1, [0]]).astype(np.uint64)
    a = oracle.build_csr(n, n, rows, cols)  # a[u, v]: edge u -> v
    A = up(ctx, a)
    At = A.transpose()
    At.build_tiles(8, 4, 1)  # 900-entry hub row: four 256-entry items
    rng = np.random.default_rng(5)
    for nf in (1, 10, 400, n):
        f = oracle.bits_from_ids(n, rng.choice(n, nf, replace=False))
        np.testing.assert_array_equal(engine.vxm(ctx, f, None, A, At, 3), oracle.vxm(a, f, None))
    e = ctx.mat_new(256, 256)  # empty (stored hypersparse): the index builds and is empty
    info = e.build_tiles(7, 1, 1)
    assert info["items"] == 0 and info["tiles"] == 2


def full_pass_cases(n, seed):
    """Frontiers / masks of the full-size passes: a random fifth of the vertices, with and without a random-half mask, and
    the dense frontier (the north-star "full-matrix pass": every stored entry is examined)."""
    rng = np.random.default_rng(seed)
    f = oracle.bits_from_ids(n, rng.choice(n, n // 5, replace=False))
    mask = oracle.bits_from_ids(n, rng.choice(n, n // 2, replace=False))
    full = oracle.bits_from_ids(n, np.arange(n))
    return [(f, None), (f, mask), (full, None), (full, mask)]


def test_tiled_full_pass_rmat22_matches_the_oracle(ctx, bench_graphs):
    """BASELINE.json configs[1] size, the layout whose roofline fraction bench.py quotes (tiled_mxv_kernel, RMAT-22): the
    next-frontier words of the LDS-tiled pass against oracle.vxm (GrB_vxm, graphblas/mod.rs:11173) over the same 65 M
    edges — random, masked and dense frontiers — and, as a second check, against the CSR pull kernel of the library."""
    A, At, a = bench_graphs(22)
    n = A.nrows
    info = At.build_tiles()
    assert info["tiles"] == 4 and info["tile_bits"] == 20
    for f, mk in full_pass_cases(n, 22):
        want = oracle.vxm(a, f, mk)
        got = engine.vxm(ctx, f, mk, A, At, 3)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(engine.vxm(ctx, f, mk, A, At, 2), want)
    full = oracle.bits_from_ids(n, np.arange(n))
    rp, _, _ = At.export_csr()
    has_in = np.diff(rp.astype(np.int64)) > 0
    np.testing.assert_array_equal(oracle.ids_from_bits(engine.vxm(ctx, full, None, A, At, 3), n), np.nonzero(has_in)[0])


# ---- blocked layout (blocked.hip): x tile AND output window in LDS -----------------------------------------------------

@pytest.fixture
def blocked(ctx):
    ctx.set_option("tiled_layout", 2)
    yield
    ctx.set_option("tiled_layout", 0)


@pytest.mark.parametrize("scale", [10, 13, 16])
def test_blocked_vxm_matches_oracle(ctx, blocked, scale):
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rng = np.random.default_rng(scale)
    A = up(ctx, a)
    At = A.transpose()
    info = At.build_tiles()
    assert info["entries"] >= a.nnz and info["entries"] % 4 == 0 and info["tile_bits"] == 18
    for nf in (1, 37, 500, n // 2, n):
        f = oracle.bits_from_ids(n, rng.choice(n, nf, replace=False))
        mask = oracle.bits_from_ids(n, rng.choice(n, n // 3, replace=False))
        for mk in (None, mask):
            want = oracle.vxm(a, f, mk)
            for _ in range(3):
                np.testing.assert_array_equal(engine.vxm(ctx, f, mk, A, At, 3), want)


def test_blocked_vxm_ragged_hubs_and_empty(ctx, blocked):
    # n not a multiple of 64 (nor of the window), empty rows, a hub row spanning several chunks, duplicates of one output
    # word inside a segment (rows 7 and 7 + 32 k share bits only across segments; row 7's 900 entries hit one word 900 times)
    n = 1000
    rows = np.concatenate([np.full(900, 7), np.arange(0, 200, 2), [999]]).astype(np.uint64)
    cols = np.concatenate([np.arange(900) + 50, np.arange(0, 200, 2) + 1, [0]]).astype(np.uint64)
    a = oracle.build_csr(n, n, rows, cols)
    A = up(ctx, a)
    At = A.transpose()
    At.build_tiles()
    rng = np.random.default_rng(5)
    for nf in (1, 10, 400, n):
        f = oracle.bits_from_ids(n, rng.choice(n, nf, replace=False))
        np.testing.assert_array_equal(engine.vxm(ctx, f, None, A, At, 3), oracle.vxm(a, f, None))
    # the same through the transposed roles (a 900-entry ROW of the matrix being streamed)
    np.testing.assert_array_equal(engine.vxm(ctx, oracle.bits_from_ids(n, np.arange(n)), None, At, A, 3),
                                  oracle.vxm(oracle.transpose(a), oracle.bits_from_ids(n, np.arange(n)), None))
    e = ctx.mat_new(256, 256)
    info = e.build_tiles()
    assert info["entries"] == 0


@pytest.mark.parametrize("scale", [22, 24, 26])
def test_blocked_full_pass_matches_the_oracle(ctx, blocked, bench_graphs, scale):
    """BASELINE sizes, the layout whose roofline fractions bench.py quotes at RMAT-24 / 26 (blocked_mxv_kernel): the blocked
    pass against oracle.vxm over the same graph (65 M / 263 M / 1.06 G edges) — random, masked and dense frontiers; the CSR
    pull kernel of the library is held to the same words as a second check."""
    A, At, a = bench_graphs(scale)
    n = A.nrows
    info = At.build_tiles()
    assert info["tile_bits"] == 18
    for f, mk in full_pass_cases(n, scale):
        want = oracle.vxm(a, f, mk)
        np.testing.assert_array_equal(engine.vxm(ctx, f, mk, A, At, 3), want)
        if scale < 26 or mk is None:
            np.testing.assert_array_equal(engine.vxm(ctx, f, mk, A, At, 2), want)
