"""GPU: fgpu_pagerank (algo.pageRank's LAGr_PageRank core, FP32) against the numpy restatement in
oracle/pagerank.py, plus the properties the reference's flow test holds (tests/flow/test_pagerank.py:40-151).

Tolerance (floating point, stated here as the task requires): BASELINE.json's north_star asks for 1e-6 relative on
PLUS_TIMES float semirings, and that is what is asserted, per score: RTOL = 1e-6, no absolute slack.  Both sides keep
FP32 vectors and accumulate every sum in FP64 before the one rounding (oracle/pagerank.py explains why that is the right
reading of an FP32 reference whose summation order is unspecified), so they differ only where an FP64 sum lands within
~1e-16 of an FP32 rounding boundary — a 1-ulp (1.2e-7) flip that the 0.85 contraction keeps below 1e-6.  The stopping
test `rdiff > tol` could still flip one iteration apart if rdiff landed within that distance of tol; the L1 distance is
then bounded by 2 * tol.  The result is reproducible bit for bit (fixed-order hub reduction, no inter-workgroup float
atomics): test_scores_are_reproducible_bit_for_bit."""
import numpy as np
import pytest

import oracle
from oracle import pagerank as opr
from falkordb_amd import engine

pytestmark = pytest.mark.gpu
U64 = np.uint64


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True, params=[0, 2], ids=["one-pass-pull", "xcd-column-ranges"])
def spmv_form(request, ctx):
    """Every test runs through both forms of the SpMV: the one-pass pull over the whole score vector (with the hub chunk
    passes) and A' split into 8 column ranges, range k gathered by XCD k out of its own L2 (pagerank.hip PrParts; the
    default picks it once the vector exceeds one L2 — forced here, the test graphs are small)."""
    ctx.set_option("pagerank_parts", request.param)
    yield
    ctx.set_option("pagerank_parts", 1)


def up(ctx, a):
    return ctx.mat_from_csr(a.nrows, a.ncols, a.rowptr, a.colidx)


RTOL = 1e-6   # north_star: "within 1e-6 rel for PLUS_TIMES float semirings"


def compare(got, it, ref, it_ref, tol=1e-4):
    assert abs(it - it_ref) <= 1, (it, it_ref)
    if it == it_ref:
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=0)
    else:
        assert float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).sum()) <= 2 * tol


def test_reference_flow_fixture_six_nodes(ctx):
    # tests/flow/test_pagerank.py:54-105: A->B->C->F->E->D->A plus E->B; B has two in-edges
    names = "ABCDEF"
    idx = {c: i for i, c in enumerate(names)}
    edges = [("A", "B"), ("B", "C"), ("C", "F"), ("F", "E"), ("E", "D"), ("D", "A"), ("E", "B")]
    a = oracle.build_csr(6, 6, np.array([idx[x] for x, _ in edges], dtype=U64),
                         np.array([idx[y] for _, y in edges], dtype=U64))
    got, it = engine.pagerank(ctx, up(ctx, a))
    ref, it_ref = opr.pagerank(a)
    compare(got, it, ref, it_ref)
    assert len(got) == 6 and (got > 0).all()
    assert abs(float(got.sum()) - 1.0) < 1e-4
    assert all(got[idx["B"]] >= got[i] for i in range(6))


def test_label_filtered_subgraph(ctx):
    # tests/flow/test_pagerank.py:107-151: A->B (Node) and S1->S2 (Special): only the Special pair is ranked
    a = oracle.build_csr(4, 4, np.array([0, 2], dtype=U64), np.array([1, 3], dtype=U64))
    active = np.array([False, False, True, True])
    got, it = engine.pagerank(ctx, up(ctx, a), active_bitmap=oracle.bits_from_ids(4, np.nonzero(active)[0]))
    ref, it_ref = opr.pagerank(a, active=active)
    compare(got, it, ref, it_ref)
    assert got[0] == 0 and got[1] == 0 and got[3] > got[2] > 0
    assert abs(float(got.sum()) - 1.0) < 1e-4


@pytest.mark.parametrize("scale", [8, 12, 16])
def test_rmat_with_sinks_matches_oracle(ctx, scale):
    a = oracle.rmat_csr(scale)
    A = up(ctx, a)
    ref, it_ref = opr.pagerank(a)
    for At in (None, A.transpose()):
        got, it = engine.pagerank(ctx, A, At)
        compare(got, it, ref, it_ref)
        assert abs(float(got.astype(np.float64).sum()) - 1.0) < 1e-4
    # other parameters: damping / tolerance / iteration cap
    got, it = engine.pagerank(ctx, A, None, None, 0.5, 1e-6, 7)
    ref, it_ref = opr.pagerank(a, 0.5, 1e-6, 7)
    assert it == it_ref == 7 or abs(it - it_ref) <= 1
    compare(got, it, ref, it_ref, 1e-6)


def test_hub_rows_and_random_label_mask(ctx):
    # a star into vertex 0 (in-degree 5999 >= HUB_DEG: the hub chunk path with float atomics), a ring, random extras
    n = 6000
    rng = np.random.default_rng(9)
    rows = np.concatenate([np.arange(1, n), np.arange(n), rng.integers(0, n, 3000)]).astype(U64)
    cols = np.concatenate([np.zeros(n - 1, dtype=np.int64), (np.arange(n) + 1) % n, rng.integers(0, n, 3000)]).astype(U64)
    a = oracle.build_csr(n, n, rows, cols)
    A = up(ctx, a)
    ref, it_ref = opr.pagerank(a)
    got, it = engine.pagerank(ctx, A)
    compare(got, it, ref, it_ref)
    assert int(np.argmax(got)) == 0
    active = rng.random(n) < 0.6
    active[0] = True
    got, it = engine.pagerank(ctx, A, None, oracle.bits_from_ids(n, np.nonzero(active)[0]))
    ref, it_ref = opr.pagerank(a, active=active)
    compare(got, it, ref, it_ref)
    assert (got[~active] == 0).all()


def test_rows_of_every_length_class(ctx):
    """Rows with 0 .. 9 in-edges from one column range, a star of 20000 in-edges into vertex 3 (thousands of entries per
    column range: whole trips of a wavefront inside one row, folded with shuffles before the LDS add), one of 700 into
    vertex 5 — three runs identical, both SpMV forms (the fixture)."""
    n = 20000
    rng = np.random.default_rng(21)
    star = np.setdiff1d(np.arange(n), [3])
    rows = [star, rng.choice(n, 700, replace=False), np.arange(n)]
    cols = [np.full(len(star), 3), np.full(700, 5), (np.arange(n) + 1) % n]
    for d in range(10):                                        # vertex 100 + d: d in-edges, all from the first range
        rows.append(np.arange(10, 10 + d))
        cols.append(np.full(d, 100 + d))
    a = oracle.build_csr(n, n, np.concatenate(rows).astype(U64), np.concatenate(cols).astype(U64))
    A = up(ctx, a)
    ref, it_ref = opr.pagerank(a)
    runs = [engine.pagerank(ctx, A) for _ in range(3)]
    compare(runs[0][0], runs[0][1], ref, it_ref)
    assert all(np.array_equal(s.view(np.uint32), runs[0][0].view(np.uint32)) and it == runs[0][1] for s, it in runs)
    assert int(np.argmax(runs[0][0])) == 3


def test_scores_are_reproducible_bit_for_bit(ctx):
    """Hub rows (in-degree >= 4096) included: three runs of the same call return identical bits, and the measured
    worst relative deviation from the oracle is reported (and far inside RTOL)."""
    a = oracle.rmat_csr(17)
    A = up(ctx, a)
    At = A.transpose()
    assert int(np.diff(oracle.transpose(a).rowptr).max()) >= 4096          # the hub chunk path really runs
    runs = [engine.pagerank(ctx, A, At) for _ in range(3)]
    assert all(it == runs[0][1] for _, it in runs)
    assert all(np.array_equal(s.view(np.uint32), runs[0][0].view(np.uint32)) for s, _ in runs)
    ref, it_ref = opr.pagerank(a)
    assert it_ref == runs[0][1]
    rel = np.abs(runs[0][0].astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)
    print(f"pagerank RMAT-17: max relative deviation from the FP64-accumulated oracle = {rel.max():.3e}")
    assert rel.max() <= RTOL


def test_empty_and_edgeless_graphs(ctx):
    a = oracle.empty(5, 5)
    got, it = engine.pagerank(ctx, up(ctx, a))
    ref, it_ref = opr.pagerank(a)
    compare(got, it, ref, it_ref)
    np.testing.assert_allclose(got, np.full(5, 0.2, dtype=np.float32), rtol=1e-6)
    got, it = engine.pagerank(ctx, up(ctx, a), None, np.zeros(1, dtype=U64))
    assert (got == 0).all() and it == 0
