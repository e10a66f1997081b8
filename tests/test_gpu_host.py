"""GPU parity of the C++ host layer (libfalkor_host.so -> libfgpu.so): Matrix<T>, VersionedMatrix, Tensor,
the Graph slice and the CondTraverse / ExpandInto / algo.BFS operators, replayed against
  * the reference's own fixtures (tests/golden/*.json, generated from /root/reference), and
  * the pure-Python oracle state machines (oracle/model.py), op for op.
Every case cites the reference test or method it mirrors (file:line relative to /root/reference)."""
import itertools
import json
import os

import numpy as np
import pytest

import oracle
from oracle import model
from falkordb_amd import host

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


PINS = gold("rust_unit_pins.json")


@pytest.fixture(scope="module")
def hctx():
    c = host.Context(0)
    yield c
    c.close()


# ---- Matrix<T> (graphblas/matrix.rs unit tests) -----------------------------------------------------
def test_matrix_build_collapses_duplicates(hctx):          # matrix.rs:1686-1695
    p = PINS["dup_collapse"]
    m = host.Matrix(hctx, host.Matrix.BOOL, p["dim"], p["dim"])
    m.build(p["rows"], p["cols"])                            # the reference's literal vectors: [1,3,1,3,1] / [2,4,2,4,2]
    m.wait()
    want = sorted(set(zip(p["rows"], p["cols"])))
    assert m.nvals() == len(want) == p["nvals"] == 2
    assert m.get(1, 2) == 1 and m.get(3, 4) == 1             # matrix.rs:1693-1694
    assert [(r, c) for r, c, _ in m.iter()] == want
    assert all(v == 1 for _, _, v in m.iter())               # bool build is iso / pattern-only (:1709-1775)


@pytest.mark.parametrize("shape", [(64, 48, 64, 96), (64, 48, 128, 48), (64, 48, 128, 96), (64, 48, 73, 55),
                                   (64, 48, 64, 48)])
def test_grown_preserves_every_entry(hctx, shape):         # matrix.rs:1617-1672
    r0, c0, r1, c1 = shape
    coords = [(i, (7 * i) % 48) for i in range(64)] + [(i, (11 * i + 3) % 48) for i in range(64)]
    vals = {rc: 1000 + k for k, rc in enumerate(coords)}     # later duplicate wins
    u = host.Matrix(hctx, host.Matrix.UINT64, r0, c0)
    u.build([c[0] for c in coords], [c[1] for c in coords], [1000 + k for k in range(len(coords))])
    g = u.grown(r1, c1)
    assert g.dims() == (r1, c1)
    assert {(r, c): v for r, c, v in g.iter()} == vals
    assert u.dims() == (r0, c0) and u.nvals() == len(vals)   # the source is untouched
    b = host.Matrix(hctx, host.Matrix.BOOL, r0, c0)
    b.build([c[0] for c in coords], [c[1] for c in coords])
    gb = b.grown(r1, c1)
    assert sorted((r, c) for r, c, _ in gb.iter()) == sorted(vals)
    assert all(v == 1 for _, _, v in gb.iter())              # bool stays pattern


def test_resize_shrink_drops_entries_past_the_new_dims(hctx):   # matrix.rs:576-598
    m = host.Matrix(hctx, host.Matrix.UINT64, 10, 10)
    m.build([0, 3, 9, 9, 5], [0, 9, 2, 9, 5], [10, 11, 12, 13, 14])
    m.resize(9, 6)
    assert m.dims() == (9, 6)
    assert m.iter() == [(0, 0, 10), (5, 5, 14)]


def test_matrix_pending_tuples_and_wait(hctx):             # matrix.rs:764-804, 1174-1184, 664-676
    m = host.Matrix(hctx, host.Matrix.UINT64, 8, 8)
    assert not m.pending() and m.nvals() == 0
    m.set(1, 2, 77)
    m.set(1, 3, 5)
    m.set(1, 2, 78)                                          # last write wins
    assert m.pending()
    assert m.get(1, 2) == 78 and not m.pending()             # extractElement finishes pending work
    m.remove(1, 3)
    m.remove(6, 6)                                           # removing an absent entry is a no-op
    assert m.get(1, 3) is None and m.get(6, 6) is None
    assert m.nvals() == 1
    d = m.dup()
    d.set(0, 0, 1)
    assert d.nvals() == 2 and m.nvals() == 1                 # dup is a deep copy (:370-385)
    with pytest.raises(host.HostError):
        m.set(8, 0, 1)                                       # GrB_INDEX_OUT_OF_BOUNDS


def test_matrix_products_match_the_oracle(hctx):           # matrix.rs:930-968, 1317-1402
    rng = np.random.default_rng(5)
    a = oracle.rmat_csr(9)
    n = a.nrows
    ar, ac = a.pairs()
    A = host.Matrix(hctx, host.Matrix.BOOL, n, n)
    A.build(ar, ac)
    fr = np.arange(40, dtype=np.uint64)
    fc = rng.integers(0, n, 40).astype(np.uint64)
    f = oracle.build_csr(40, n, fr, fc)
    F = host.Matrix(hctx, host.Matrix.BOOL, 40, n)
    F.build(fr, fc)
    F.lmxm(A)
    want, _ = oracle.mxm(f, a)
    assert [(r, c) for r, c, _ in F.iter()] == sorted(want.to_set())
    # rmxm: self = b * self
    G = host.Matrix(hctx, host.Matrix.BOOL, n, n)
    G.build(ar, ac)
    F2 = host.Matrix(hctx, host.Matrix.BOOL, 40, n)
    F2.build(fr, fc)
    G.rmxm(F2)
    assert [(r, c) for r, c, _ in G.iter()] == sorted(want.to_set())
    # delta_lmxm with non-empty dp / dm
    dm_idx = rng.choice(a.nnz, 200, replace=False)
    dmr, dmc = ar[dm_idx], ac[dm_idx]
    dpr, dpc = rng.integers(0, n, 300).astype(np.uint64), rng.integers(0, n, 300).astype(np.uint64)
    DP = host.Matrix(hctx, host.Matrix.BOOL, n, n); DP.build(dpr, dpc)
    DM = host.Matrix(hctx, host.Matrix.BOOL, n, n); DM.build(dmr, dmc)
    F3 = host.Matrix(hctx, host.Matrix.BOOL, 40, n); F3.build(fr, fc)
    F3.delta_lmxm(A, DP, DM)
    w, _ = oracle.delta_lmxm(f, a, oracle.build_csr(n, n, dpr, dpc), oracle.build_csr(n, n, dmr, dmc))
    assert [(r, c) for r, c, _ in F3.iter()] == sorted(w.to_set())
    assert A.intersection_nvals(DM) == len(set(zip(dmr.tolist(), dmc.tolist())))


def test_transpose_carries_values(hctx):                   # matrix.rs:633-662
    rng = np.random.default_rng(11)
    n = 6000
    r = rng.integers(0, n, n).astype(np.uint64)
    c = rng.integers(0, 50, n).astype(np.uint64)
    v = rng.integers(0, 1 << 62, n).astype(np.uint64)
    m = host.Matrix(hctx, host.Matrix.UINT64, n, 50)
    m.build(r, c, v)
    want = {}
    for a, b, x in zip(r.tolist(), c.tolist(), v.tolist()):
        want[(b, a)] = x                                     # last duplicate wins
    t = m.transpose()
    assert t.dims() == (50, n)
    assert {(a, b): x for a, b, x in t.iter()} == want


# ---- VersionedMatrix (graphblas/versioned_matrix.rs unit tests) -----------------------------------------
def lcg_next(state):                                       # versioned_matrix.rs:1380-1385
    p = PINS["lcg_model"]
    state = (state * p["mul"] + p["add"]) & ((1 << 64) - 1)
    return state, state >> p["shift"]


def check_invariants(v: host.VersionedMatrix, ref: set):   # versioned_matrix.rs:1345-1377
    it = v.iter()
    assert it == sorted(ref)
    assert v.nvals() == len(ref)
    ext = v.extract()
    assert [(r, c) for r, c, _ in ext.iter()] == sorted(ref)
    st = v.state()
    assert st["m"] + st["dp"] - st["dm"] == len(ref)         # dp & m = {}, dm subset of m


def test_delta_invariants_hold_across_mutation_sequences(hctx):   # versioned_matrix.rs:1399-1472
    p = PINS["lcg_model"]
    dim = p["dim"]
    v = host.VersionedMatrix(hctx, dim, dim)
    o = model.VersionedMatrix(dim, dim)
    ref = set()
    rng = p["seed"]
    key = lambda r: ((r % 24) * 7, (r // 24 % 24) * 11)

    def rnd():
        nonlocal rng
        rng, out = lcg_next(rng)
        return out

    folded = False
    for step in range(p["steps"]):
        op = rnd() % 16
        if op == 0:
            batch = [key(rnd()) for _ in range(16)]
            v.set_all(batch, new=False); o.set_all(batch, new=False)
            ref.update(batch)
        elif op == 1:
            batch = sorted({key(rnd()) for _ in range(16)})
            v.remove_mask(batch); o.remove_mask(batch)
            ref.difference_update(batch)
        elif op == 2:
            v = v.dup(); o = o.dup()
        elif op == 3:
            v.wait(); o.wait()
        elif op == 4:
            v.fold_oversized(); o.fold_oversized()
        elif 5 <= op <= 8:
            k = key(rnd())
            v.remove(*k); o.remove(*k)
            ref.discard(k)
        else:
            k = key(rnd())
            v.set(*k); o.set(k[0], k[1], True)
            ref.add(k)
        if step % p["check_stride"] == 0:
            check_invariants(v, ref)                         # reads go through wait() (fold decisions latch there)
            assert o.iter() == sorted(ref) and o.nvals() == len(ref)   # the same reads on the oracle
            # layer for layer the same state as the oracle state machine (fold decisions included)
            o.wait_all()
            st = v.state()
            assert (st["m"], st["dp"], st["dm"]) == (len(o.m), len(o.dp.layer), len(o.dm.layer)), step
            assert st["needs_flush"] == o.needs_flush
            folded = folded or st["m"] > 0
    check_invariants(v, ref)
    assert folded, "no fold ever happened: the `m` branches of set/remove were never taken"


def test_folded_entry_deleted_and_re_added_stays_out_of_dp(hctx):  # versioned_matrix.rs:1481-1523
    p = PINS["refold_probe"]
    DIM, filler = p["dim"], p["filler"]
    v = host.VersionedMatrix(hctx, DIM, DIM)
    v.set_all([(i % DIM, (i // DIM + 1) % DIM) for i in range(filler)], new=False)
    probe = tuple(p["probe"])
    v.set(*probe)
    v = v.dup()
    v.set(*p["trigger"])
    v.wait()
    st = v.state()
    assert st["m"] == filler + 1 and st["dp"] == 1           # the dup's latched fold ran before the first mutation
    v.remove(*probe)
    assert v.get(*probe) is None
    st = v.state()
    assert st["dm"] == 1
    v.set(*probe)
    assert v.get(*probe) is True
    st = v.state()
    assert st["dm"] == 0 and st["dp"] == 1                   # un-deleted, NOT re-added to dp
    assert v.nvals() == p["final_nvals"] == filler + 2


def test_versioned_matrix_transpose_and_from_matrix(hctx):   # versioned_matrix.rs:877-890, 1070-1079
    rows, cols = [0, 0, 3, 5, 5], [1, 4, 3, 0, 2]
    v = host.VersionedMatrix.from_coo(hctx, 6, 7, rows, cols)
    v.set(2, 6)
    v.remove(0, 4)
    live = (set(zip(rows, cols)) | {(2, 6)}) - {(0, 4)}
    assert v.iter() == sorted(live)
    t = v.transpose()
    assert t.iter() == sorted((c, r) for r, c in live)
    assert t.state()["dp"] == 1 and t.state()["dm"] == 1


# ---- Tensor state machine (graphblas/tensor.rs:72-108; tests at :1340-1669) ---------------------------------
class TensorModel:
    """(src, dst) -> sorted edge ids; the observable behaviour every layer state must reproduce."""

    def __init__(self):
        self.pairs = {}

    def add(self, s, d, e):
        self.pairs.setdefault((s, d), [])
        if e not in self.pairs[(s, d)]:
            self.pairs[(s, d)].append(e)
            self.pairs[(s, d)].sort()

    def remove(self, s, d, e):
        ids = self.pairs.get((s, d), [])
        if e in ids:
            ids.remove(e)
            if not ids:
                del self.pairs[(s, d)]

    def edges(self):
        return sorted((s, d, e) for (s, d), ids in self.pairs.items() for e in ids)


def check_tensor(g: host.Graph, t, mdl: TensorModel, n):
    assert sorted(g.tensor_iter_edges(t)) == mdl.edges()
    assert g.tensor_edge_count(t) == len(mdl.edges())
    st = g.tensor_state(t)
    assert st["multi_pairs"] == sum(1 for ids in mdl.pairs.values() if len(ids) > 1)
    assert st["mt"] == len(mdl.pairs)                        # mt mirrors the effective forward structure
    for (s, d), ids in mdl.pairs.items():
        assert g.tensor_get(t, s, d) == ids


@pytest.mark.parametrize("commits", list(itertools.product([False, True], repeat=5)),
                         ids=lambda c: "".join("C" if x else "-" for x in c))
def test_multiple_edges_lifecycle_ids(hctx, commits):      # tests/flow/test_multiple_edges.py:11-96
    steps = gold("multiple_edges.json")["steps"]
    g = host.Graph(hctx, 2)
    t = g.add_type("R")
    mdl = TensorModel()
    assert g.tensor_get(t, 0, 1) == steps[0]["ids"]
    for st, commit in zip(steps[1:], commits):
        if st["op"] == "add":
            g.create_edge(t, 0, 1, st["id"]); mdl.add(0, 1, st["id"])
        else:
            g.delete_edge(t, 0, 1, st["id"]); mdl.remove(0, 1, st["id"])
        assert g.tensor_get(t, 0, 1) == st["ids"], (st, commits)
        check_tensor(g, t, mdl, 2)
        # the pair is in the adjacency matrix exactly while it holds an edge
        rows = g.cond_traverse_batch(host.cond_spec(hops=[([], [])]), [0])[0]
        assert [d for _, d in rows] == ([1] if st["ids"] else [])
        if commit:
            g.commit()
            assert g.tensor_get(t, 0, 1) == st["ids"], ("after commit", st, commits)
            check_tensor(g, t, mdl, 2)


def test_tensor_random_walk_matches_the_pair_model(hctx):  # tensor.rs:1340-1669 (promote / demote / cancel / fold)
    rng = np.random.default_rng(2024)
    n = 12
    g = host.Graph(hctx, n)
    t = g.add_type("R")
    mdl = TensorModel()
    next_id = 0
    live = []
    for step in range(400):
        op = rng.integers(0, 10)
        if op < 5 or not live:
            k = int(rng.integers(1, 5))
            s = rng.integers(0, 4, k).tolist()
            d = rng.integers(0, 4, k).tolist()
            ids = list(range(next_id, next_id + k))
            next_id += k
            g.create_edges(t, s, d, ids)
            for a, b, e in zip(s, d, ids):
                mdl.add(a, b, e)
                live.append((a, b, e))
        elif op < 9:
            a, b, e = live.pop(int(rng.integers(0, len(live))))
            g.delete_edge(t, a, b, e)
            mdl.remove(a, b, e)
        else:
            g.commit()
        if step % 7 == 0:
            check_tensor(g, t, mdl, n)
    check_tensor(g, t, mdl, n)
    # the adjacency matrix tracks pairs with >= 1 edge
    rows, _, _ = g.cond_traverse_batch(host.cond_spec(hops=[([], [])]), list(range(n)))
    assert sorted(rows) == sorted(mdl.pairs)


def test_tensor_rust_unit_pins(hctx):
    """The #[test]s of tensor.rs:1340-1669, transcribed one for one (tests/golden/rust_unit_pins.json lists them) and
    run against the C++ Tensor on the device.  Each block carries its reference test's name and assertions."""
    T = PINS["tensor_unit_tests"]
    M = host.Tensor.MULTI_EDGE

    # multi_pairs_after_within_batch_duplicates (:1340-1381): one batch, each pair repeated `dup` times consecutively
    for pairs, dup in T["multi_pairs_after_within_batch_duplicates"]["cases"]:
        t = host.Tensor(hctx, pairs + 1, pairs + 1)
        srcs = np.repeat(np.arange(pairs), dup)
        t.set_all_from_slices(srcs, srcs + 1, np.arange(pairs * dup))
        t.wait_fwd()
        sentinels = sum(t.eff_get(i, i + 1) == M for i in range(pairs))
        edges = sum(len(t.get(i, i + 1)) for i in range(0, pairs, max(1, pairs // 64))) * 1
        st = t.state()
        assert st["multi_pairs"] == sentinels == pairs                       # "multi_pairs disagrees (within-batch dups)"
        assert st["edge_count"] == pairs * dup                               # "edge_count disagrees with a full scan"
        assert edges == dup * len(range(0, pairs, max(1, pairs // 64)))
        assert t.get(0, 1) == list(range(dup))

    # multi_pairs_matches_the_sentinel_count (:1386-1425): `dup` edges per pair, inserted as separate batches
    for pairs, dup in T["multi_pairs_matches_the_sentinel_count"]["cases"]:
        t = host.Tensor(hctx, pairs + 1, pairs + 1)
        srcs = np.arange(pairs)
        for rnd in range(dup):
            t.set_all_from_slices(srcs, srcs + 1, rnd * pairs + srcs)
        t.wait_fwd()
        probe = range(0, pairs, max(1, pairs // 128))
        assert all(t.eff_get(i, i + 1) == M for i in probe)
        st = t.state()
        assert st["multi_pairs"] == pairs and st["edge_count"] == pairs * dup and st["me_nvals"] == pairs * dup
        assert t.get(3, 4) == [3 + r * pairs for r in range(dup)]

    # bulk_remove_and_extract_edge_id_zero (:1427-1460)
    c = T["bulk_remove_and_extract_edge_id_zero"]
    n = c["n"]
    t = host.Tensor(hctx, n + 1, n + 1)
    srcs = np.arange(n)
    t.set_all_from_slices(srcs, srcs + 1, srcs)
    t = t.dup()
    t.flush()
    assert t.m_get(0, 1) == 0                                                # "edge id 0 not folded into base"
    t.remove_all([tuple(r) for r in c["remove"]])
    assert t.get(0, 1) == [] and t.get(5, 6) == []                           # "edge id 0 / 5 still readable"
    assert t.extract_contains(1, 2)                                          # "unrelated live pair (1,2) disappeared"
    assert not t.extract_contains(5, 6)                                      # "control pair (5,6) not deleted"
    assert not t.extract_contains(0, 1)                                      # edge id 0 must not be typecast to false in dm

    # resize_leaves_base_materialized (:1478-1501)
    c = T["resize_leaves_base_materialized"]
    n = c["n"]
    t = host.Tensor(hctx, n + 1, n + 1)
    t.set_all_from_slices(srcs, srcs + 1, srcs)
    t.flush()
    t.wait()
    t.resize(c["resize_to"], c["resize_to"])
    assert t.state()["m_pending"] == 0                                       # "resize left the committed base pending"
    t.wait_fwd()
    assert t.get(0, 1) == [0]                                                # "edge lost across resize"

    # deleting_everything_folds_the_tombstones_away (:1509-1535)
    n = T["deleting_everything_folds_the_tombstones_away"]["n"]
    t = host.Tensor(hctx, n + 1, n + 1)
    t.set_all_from_slices(srcs, srcs + 1, srcs)
    t = t.dup()
    t.flush()
    t.wait_fwd()
    assert t.state()["m"] == n                                               # "adds did not fold into the base"
    t = t.dup()
    t.remove_all([(i, i, i + 1) for i in range(n)])
    t.fold_oversized()
    st = t.state()
    assert st["m"] == 0 and st["dm"] == 0 and t.get(0, 1) == []              # base / tombstones folded away

    N = T["FOLDABLE"]

    def committed_pairs(k):                                                  # (:1543-1556)
        t = host.Tensor(hctx, k + 1, k + 1)
        s2 = np.repeat(np.arange(k), 2)
        t.set_all_from_slices(s2, s2 + 1, np.arange(2 * k))
        t.fold_oversized()
        t.wait()
        assert t.state()["m"] == k                                           # "sentinels not folded into the base"
        return t.dup()

    # batch_demote_leaves_every_survivor_inline (:1563-1584)
    t = committed_pairs(N)
    emptied = t.remove_all([(2 * i + 1, i, i + 1) for i in range(N)])
    st = t.state()
    assert emptied == [] and st["edge_count"] == N and st["multi_pairs"] == 0 and st["me_nvals"] == 0
    assert all(t.get(i, i + 1) == [2 * i] for i in range(N))                 # "pair lost its surviving edge"

    # batch_can_demote_and_then_empty_the_same_pair (:1590-1630)
    t = committed_pairs(N)
    rels = []
    for i in range(N):
        rels += [(2 * i + 1, i, i + 1), (2 * i + 1, i, i + 1), (7 * N + i, i, i + 1), (2 * i, i, i + 1), (2 * i, i, i + 1)]
    emptied = sorted(t.remove_all(rels))
    assert emptied == [(i, i + 1) for i in range(N)]                         # every pair emptied exactly once
    st = t.state()
    assert st["edge_count"] == 0 and st["multi_pairs"] == 0 and st["me_nvals"] == 0 and st["mt"] == 0
    assert all(t.get(i, i + 1) == [] for i in range(N))

    # demoting_to_the_committed_value_cancels_to_clean (:1636-1668)
    t = host.Tensor(hctx, N + 1, N + 1)
    s1 = np.arange(N)
    t.set_all_from_slices(s1, s1 + 1, s1 + 1)
    t.fold_oversized()
    t.wait()
    assert t.m_get(0, 1) == 1                                                # "single edge not committed"
    t = t.dup()
    t.set_all_from_slices([0], [1], [9])
    emptied = t.remove_all([(9, 0, 1)])
    assert emptied == []
    t.wait()
    st = t.state()
    assert st["dp"] == 0 and st["dm"] == 0 and st["me_nvals"] == 0 and st["edge_count"] == N
    assert t.get(0, 1) == [1]                                                # "committed edge lost"


def test_tensor_v19_payload_with_multi_edges(hctx):
    """Encode<19> / Decode<19> for Tensor (tensor.rs:1053-1209): a tensor with single edges, multi-edge pairs (committed
    and pending), a deleted pair and edge id 0 survives encode -> decode; the payload's framing is checked field by field
    against the layout the reference writes; a foreign (GraphBLAS-serialised) id-list blob is refused, not guessed at."""
    from test_host_cpu import _frame_u
    n = 300
    t = host.Tensor(hctx, n, n)
    s = np.arange(100)
    t.set_all_from_slices(s, s + 1, s)                       # ids 0..99, one per pair (edge id 0 included)
    t.set_all_from_slices([5, 5, 9], [6, 6, 10], [500, 501, 502])   # (5,6) -> 3 edges, (9,10) -> 2 edges
    t.fold_oversized()
    t = t.dup()
    t.set_all_from_slices([200, 200], [201, 201], [900, 901])       # a pending multi pair
    t.remove_all([(42, 42, 43)])                                     # a tombstone
    payload = t.encode()
    back, used = host.Tensor.decode(hctx, payload)
    assert used == len(payload)
    st, sb = t.state(), back.state()
    assert sb["edge_count"] == st["edge_count"] == 100 - 1 + 3 + 2
    assert sb["multi_pairs"] == st["multi_pairs"] == 3 and sb["me_nvals"] == st["me_nvals"] == 3 + 2 + 2
    assert (sb["dp"], sb["dm"]) == (0, 0)                            # deltas are folded into the on-disk base (:1083-1094)
    assert sb["mt"] == st["mt"]                                      # backward matrix rebuilt
    for (a, b) in [(0, 1), (5, 6), (9, 10), (200, 201), (42, 43), (7, 8)]:
        assert back.get(a, b) == t.get(a, b)
    assert back.get(0, 1) == [0] and back.get(5, 6) == [5, 500, 501] and back.get(42, 43) == []
    assert back.eff_get(5, 6) == host.Tensor.MULTI_EDGE
    assert back.encode() == payload                                  # canonical
    # framing: three containers, then total, base count, (src, dst, blob)*, 0
    d = host.container_parse(payload)
    assert d["valued"] and d["nvals"] == 100 - 1 + 1 and (d["nrows"], d["ncols"]) == (n, n)   # pairs: one deleted, one new
    msb = 1 << 63
    multi_vals = sorted(int(x) & ~msb for x in d["x"] if int(x) & msb)
    assert multi_vals == [2, 2, 3]                                   # count | MSB for the three multi pairs
    off = d["consumed"]
    d2 = host.container_parse(payload[off:]); off += d2["consumed"]
    d3 = host.container_parse(payload[off:]); off += d3["consumed"]
    assert d2["nvals"] == 0 and d3["nvals"] == 0
    assert d2["valued"] and d3["valued"]      # both deltas are the SAME empty Matrix<u64> the reference writes (tensor.rs:1093-1094)
    rd = lambda o: int.from_bytes(payload[o:o + 8], "little")
    assert rd(off) == sb["edge_count"] and rd(off + 8) == 3          # total edges, multi pairs in the base group
    assert (rd(off + 16), rd(off + 24)) == (5, 6)                    # first pair in (row, col) order
    assert payload[-8:] == _frame_u(0)                               # empty delta-plus group
    # the u64 before the groups is only tested for > 0 by the reference's decoder (tensor.rs:1169-1170): a writer that
    # stores the number of TENSORS there (3) instead of the number of edges decodes to the same tensor
    alt, _ = host.Tensor.decode(hctx, payload[:off] + _frame_u(3) + payload[off + 8:])
    assert alt.state()["edge_count"] == sb["edge_count"] and alt.get(5, 6) == [5, 500, 501] and alt.get(0, 1) == [0]
    # a pair flagged multi-edge whose id list is missing from the tensor section (the first of the three triples dropped,
    # the group count lowered): the reference's decoder never cross-checks the two sections (tensor.rs:1169-1186) — the
    # payload decodes, the pair keeps its MULTI_EDGE inline value and yields no ids, everything else is intact
    first_len = rd(off + 32)
    holed, used_h = host.Tensor.decode(hctx, payload[:off + 8] + _frame_u(2) + payload[off + 40 + first_len:])
    assert used_h == len(payload) - (24 + first_len)
    assert holed.eff_get(5, 6) == host.Tensor.MULTI_EDGE and holed.get(5, 6) == []
    assert holed.get(9, 10) == t.get(9, 10) and holed.get(200, 201) == [900, 901] and holed.get(0, 1) == [0]
    # an empty tensor: three empty containers and a zero count, nothing else
    e = host.Tensor(hctx, 10, 10)
    pe = e.encode()
    be, used = host.Tensor.decode(hctx, pe)
    assert used == len(pe) and be.state()["edge_count"] == 0 and pe[-8:] == _frame_u(0)
    # a blob that is not this library's plain list (what GxB_Vector_serialize writes starts with its own size)
    blob_at = off + 32
    blob_len = rd(blob_at)
    foreign = payload[:blob_at + 8] + blob_len.to_bytes(8, "little") + payload[blob_at + 16:]
    with pytest.raises(host.HostError) as ei:
        host.Tensor.decode(hctx, foreign)
    assert "GxB_Vector" in str(ei.value)


def test_bulk_delete_folds_tombstones_at_commit(hctx):     # tensor.rs:1590-1669 (delete-all folds tombstones)
    n = 2000
    g = host.Graph(hctx, n)
    t = g.add_type("R")
    s = np.arange(n - 1)
    g.create_edges(t, s, s + 1, np.arange(n - 1))
    g.commit()                                               # dp dominates an empty base: folded into m
    st = g.tensor_state(t)
    assert (st["m"], st["dp"], st["dm"]) == (n - 1, 0, 0)
    for i in range(0, n - 1):
        g.delete_edge(t, i, i + 1, i)
    st = g.tensor_state(t)
    assert (st["m"], st["dp"], st["dm"]) == (n - 1, 0, n - 1)
    assert g.tensor_edge_count(t) == 0
    g.commit()                                               # escape hatch: 2|dm| >= |m|
    st = g.tensor_state(t)
    assert (st["m"], st["dp"], st["dm"]) == (0, 0, 0)
    assert g.tensor_iter_edges(t) == []


# ---- social graph known answers through the operators (tests/flow/social) ----------------------------------
def social(hctx, commit):
    s = gold("social.json")
    names = [p["name"] for p in s["persons"]] + s["countries"]
    ids = {n: i for i, n in enumerate(names)}
    g = host.Graph(hctx, len(names))
    lp, lc = g.add_label("person"), g.add_label("country")
    for p in s["persons"]:
        g.label_node(ids[p["name"]], lp)
    for c in s["countries"]:
        g.label_node(ids[c], lc)
    tf, tv = g.add_type("friend"), g.add_type("visited")
    eid = 0
    for a, b, _purpose in s["visits"]:
        g.create_edge(tv, ids[a], ids[b], eid); eid += 1
    if commit:
        g.commit()
    for a, b in s["friends"]:
        g.create_edge(tf, ids[a], ids[b], eid); eid += 1
    if commit:
        g.commit()
    return s, g, ids, names


@pytest.mark.parametrize("commit", [False, True], ids=["pending-deltas", "committed"])
def test_social_queries(hctx, commit):                     # social_queries.py:57-80
    s, g, ids, names = social(hctx, commit)
    roi = ids["Roi Lipman"]
    one = host.cond_spec(src_labels=["person"], hops=[(["friend"], ["person"])])
    rows, nulls, _ = g.cond_traverse_batch(one, [roi])
    assert sorted(names[d] for _, d in rows) == sorted(r[0] for r in s["queries"]["my_friends_query"]["expected"])
    mids = [d for _, d in rows]
    rows2, _, _ = g.cond_traverse_batch(host.cond_spec(hops=[(["friend"], ["person"])]), mids)
    fof = [names[d] for _, d in rows2]
    assert sorted(fof) == sorted(r[0] for r in s["queries"]["friends_of_friends_query"]["expected"])
    # anonymous unlabeled intermediate: ONE fused CondTraverse with a chain (fuse_anonymous_traverse.rs:83-188)
    fused = host.cond_spec(src_labels=["person"], hops=[(["friend"], []), (["friend"], ["person"])])
    rows3, _, flops = g.cond_traverse_batch(fused, [roi])
    assert sorted(names[d] for _, d in rows3) == sorted(set(fof))
    assert flops > 0
    # the same operator obtained from the plan: MATCH (a:person)-[:friend]->()-[:friend]->(c:person) — two
    # CondTraverse nodes that fuse_anonymous_traverse merges, lowered to the runtime spec
    plan = [{"id": 0, "parent": -1, "kind": "X", "name": "Project", "refs": ["c"]},
            {"id": 1, "parent": 0, "kind": "CT", "rel": {"alias": "_anon_e2", "from": {"alias": "_anon_1", "labels": []},
                                                         "to": {"alias": "c", "labels": ["person"]}, "types": ["friend"]}},
            {"id": 2, "parent": 1, "kind": "CT", "rel": {"alias": "_anon_e1", "from": {"alias": "a", "labels": ["person"]},
                                                         "to": {"alias": "_anon_1", "labels": []}, "types": ["friend"]}},
            {"id": 3, "parent": 2, "kind": "X", "name": "NodeByLabelScan", "refs": []}]
    after, spec = host.plan_fuse(plan, lower_id=1)
    assert sum(o["kind"] == "CT" for o in after) == 1
    rows5, _, _ = g.cond_traverse_batch(spec, [roi])
    assert rows5 == rows3
    # visited Netherlands and single
    attrs = {p["name"]: p for p in s["persons"]}
    singles = [ids[n] for n in fof if attrs[n]["status"] == "single"]
    rows4, _, _ = g.cond_traverse_batch(host.cond_spec(hops=[(["visited"], ["country"])]), singles)
    got = sorted({names[singles[i]] for i, d in rows4 if names[d] == "Netherlands"})
    assert got == [r[0] for r in s["queries"]["friends_of_friends_visited_netherlands_and_single_query"]["expected"]]
    # relation counts
    want = dict(map(tuple, s["queries"]["relation_type_counts"]["expected"]))
    assert g.tensor_edge_count(0) == want["friend"] and g.tensor_edge_count(1) == want["visited"]


# ---- imdb fixture known answers through the operators (tests/flow/imdb, imdb_queries.py) ------------------------
@pytest.mark.parametrize("commit", [False, True], ids=["pending-deltas", "committed"])
def test_imdb_known_answers(hctx, commit):
    d = gold("imdb.json")
    names = d["actors"] + d["movies"]
    ida = {n: i for i, n in enumerate(d["actors"])}
    idm = {n: len(d["actors"]) + i for i, n in enumerate(d["movies"])}
    g = host.Graph(hctx, len(names))
    la, lm = g.add_label("actor"), g.add_label("movie")
    for n in d["actors"]:
        g.label_node(ida[n], la)
    for n in d["movies"]:
        g.label_node(idm[n], lm)
    t = g.add_type("act")
    g.create_edges(t, [ida[a] for a, _ in d["act"]], [idm[m] for _, m in d["act"]], list(range(len(d["act"]))))
    if commit:
        g.commit()
    q = d["queries"]
    fwd = host.cond_spec(src_labels=["actor"], hops=[(["act"], ["movie"])])
    # (a:actor)-[:act]->(m) with only m bound: the per-row path over the transposed structure (cond_traverse.rs:840-870)
    cast_of = lambda m: sorted({f for f, _, _ in g.cond_traverse_row(fwd, to_id=m)})
    rows, _, _ = g.cond_traverse_batch(fwd, [ida["Nicolas Cage"]])
    got = sorted([names[a], names[m]] for _, m in rows for a in cast_of(m))
    assert got == q["actors_played_with_nicolas_cage_query"]["expected"]
    assert sorted(names[a] for a in cast_of(idm["Straight Outta Compton"])) == \
        sorted(r[0] for r in q["actors_played_in_movie_straight_outta_compton_query"]["expected"])
    rows, _, _ = g.cond_traverse_batch(fwd, [ida["Cameron Diaz"]])
    assert [["Cameron Diaz", len(rows)]] == q["how_many_movies_cameron_diaz_played_query"]["expected"]
    gbh = idm["The Grand Budapest Hotel"]
    cast = cast_of(gbh)
    opt = host.cond_spec(hops=[(["act"], ["movie"])], optional=True)
    rows, nulls, _ = g.cond_traverse_batch(opt, cast)
    assert nulls == []                                        # everyone played in GBH itself
    got = [[names[cast[i]], names[m]] for i, m in rows if m != gbh]
    matched = {i for i, m in rows if m != gbh}
    got += [[names[cast[i]], None] for i in range(len(cast)) if i not in matched]   # WHERE m <> h leaves the NULL row
    key = lambda r: (r[0], r[1] or "")
    assert sorted(got, key=key) == sorted(q["grand_budapest_hotel_cast_and_their_other_roles"]["expected"], key=key)
    assert g.tensor_edge_count(t) == len(d["act"])


# ---- BASELINE config 1: LDBC-SNB-SF0.1-shaped 2-hop MATCH through the whole stack ----------------------------
def test_ldbc_shaped_two_hop_match_count(hctx):
    """`MATCH (a:Person)-[:KNOWS]->()-[:KNOWS]->(c) RETURN count(c)` (bench row "two-hop",
    bench/src/falkorbench/queries.py:184) on a stand-in with SF0.1's shape (the LDBC dataset is not available
    offline): ~1.5 k Person nodes among other nodes, ~18 k directed KNOWS edges with a power-law
    degree distribution, plus a second relationship type that must not leak into the result.  The fused
    CondTraverse chain runs through libfalkor_host -> fgpu_expand in batches of 1024 rows (batch.rs:81)."""
    rng = np.random.default_rng(2026)
    n_person, n_other = 1536, 512
    n = n_person + n_other
    e = oracle.rmat_edges(11, edge_factor=9, seed=0xD0C)       # 2048-vertex R-MAT, ~18 k raw edges
    keep = (e[0] < n_person) & (e[1] < n_person)
    ks, kd = e[0][keep], e[1][keep]
    g = host.Graph(hctx, n)
    og = model.Graph(n)
    lp = g.add_label("Person"); og.add_label("Person")
    for v in range(n_person):
        g.label_node(v, lp); og.node_labels.add((v, lp))
    tk, tl = g.add_type("KNOWS"), g.add_type("LIKES")
    og.add_type("KNOWS"); og.add_type("LIKES")
    g.create_edges(tk, ks, kd, np.arange(len(ks)))
    ls, ld = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    g.create_edges(tl, ls, ld, np.arange(len(ks), len(ks) + 4000))
    g.commit()
    pairs = {}
    for s_, d_, i_ in zip(ks.tolist(), kd.tolist(), range(len(ks))):
        pairs.setdefault((s_, d_), []).append(i_)
    for p_, es in pairs.items():
        og.tensors[0].m[p_] = es[0] if len(es) == 1 else model.MULTI_EDGE
        if len(es) > 1:
            og.tensors[0].me[p_] = es
        og.adjacency.m.add(p_)
    spec = host.cond_spec(src_labels=["Person"], hops=[(["KNOWS"], []), (["KNOWS"], [])])
    sources = list(range(n))                                     # the label scan feeds every node id
    count, want = 0, 0
    for b in range(0, n, 1024):
        batch = sources[b:b + 1024]
        (rows, dest, _), nulls, flops = g.cond_traverse_batch(spec, batch, as_arrays=True)
        count += len(dest)
        ref = model.expand_batch(og, batch, ["KNOWS"], src_labels=["Person"], chain=[(["KNOWS"], [])])[0]
        want += len(ref)
        assert list(zip(rows.tolist(), dest.tolist())) == ref
    assert count == want > 10000


# ---- algo.BFS known answers (tests/flow/test_bfs.py:10-25, 63-213) ------------------------------------------
def bfs5(hctx, commit):
    b = gold("bfs5.json")
    ids = {n: i for i, n in enumerate(b["nodes"])}
    g = host.Graph(hctx, len(ids))
    for eid, (s, d, t) in enumerate(b["edges"]):
        g.create_edge(g.add_type(t), ids[s], ids[d], eid)
    if commit:
        g.commit()
    return b, g, ids


@pytest.mark.parametrize("commit", [False, True], ids=["pending-deltas", "committed"])
def test_algo_bfs_known_answers(hctx, commit):
    b, g, ids = bfs5(hctx, commit)
    names = b["nodes"]
    dst_of = {eid: d for eid, (s, d, t) in enumerate(b["edges"])}
    for case in b["cases"]:
        res = g.algo_bfs(ids[case["src"]], case["depth"], case["type"], want_edges="edges_dst" in case)
        if case["nodes"] is None:
            assert res is None, case
            continue
        nodes, edges = res
        assert nodes == sorted(nodes)                        # ascending index of the level vector
        assert sorted(names[v] for v in nodes) == sorted(case["nodes"]), case
        if "edges_dst" in case:
            assert sorted(dst_of[e] for e in edges) == sorted(case["edges_dst"]), case
    assert g.algo_bfs(None) is None                          # NULL source: no row
    g.delete_node(ids[b["cases"][0]["src"]])
    with pytest.raises(host.HostError):                      # deleted source: error (algo_procedures.rs:1058-1064)
        g.algo_bfs(ids[b["cases"][0]["src"]])


# ---- ExpandInto known answers (tests/flow/test_expand_into.py:17-95) ----------------------------------------
@pytest.mark.parametrize("case", gold("expand_into.json")["cases"], ids=lambda c: c["name"])
@pytest.mark.parametrize("batched", [False, True], ids=["per-row", "batched"])
def test_expand_into_known_answers(hctx, case, batched):
    g = host.Graph(hctx, case["nodes"])
    for eid, (s, d, t) in enumerate(case["edges"]):
        g.create_edge(g.add_type(t), s, d, eid)
    a, b = case["a"], case["b"]
    if "count_named_edge" in case:
        named = g.expand_into(case["types"], [a], [b], emit_relationship=True, batched=batched)
        anon = g.expand_into(case["types"], [a], [b], emit_relationship=False, batched=batched)
        assert len(named) == case["count_named_edge"] and len(anon) == case["count_pairs"]
    if "two_hop_rows" in case:
        spec = host.cond_spec(hops=[(["R"], []), ([], [])])
        rows, _, _ = g.cond_traverse_batch(spec, [a], to_bound=[b])
        assert len(rows) == case["two_hop_rows"]
        if "expected_count" in case:
            assert case["varlen_trails"] * len(rows) == case["expected_count"]


# ---- operators on a random R-MAT graph with labels and dirty deltas vs the oracle --------------------------
def build_random(hctx, scale=9, seed=3):
    rng = np.random.default_rng(seed)
    a = oracle.rmat_csr(scale)
    n = a.nrows
    er, ec = a.pairs()
    g = host.Graph(hctx, n)
    og = model.Graph(n)
    labels = ["P", "Q"]
    lids = [g.add_label(l) for l in labels]
    for l in labels:
        og.add_label(l)
    for v in range(n):
        for k, lid in enumerate(lids):
            if (oracle.mix64(np.uint64(v * 7 + k))) % np.uint64(3 + k) == 0:
                g.label_node(v, lid)
                og.node_labels.add((v, lid))
    types = ["A", "B"]
    tids = [g.add_type(t) for t in types]
    for t in types:
        og.add_type(t)
    split = rng.integers(0, 2, len(er))
    eid = 0
    per_type = {0: {}, 1: {}}
    for k in (0, 1):
        sel = np.nonzero(split == k)[0]
        s, d = er[sel], ec[sel]
        ids = np.arange(eid, eid + len(sel))
        eid += len(sel)
        g.create_edges(tids[k], s, d, ids)
        for x, y, e in zip(s.tolist(), d.tolist(), ids.tolist()):
            per_type[k][(x, y)] = [e]
    g.commit()   # everything so far lands in the committed bases (dp dominates the empty base)
    # a second transaction: extra parallel edges (promotions), fresh pairs (dp) and deletions (dm)
    for k in (0, 1):
        keys = list(per_type[k])
        for j in rng.choice(len(keys), 60, replace=False):
            x, y = keys[j]
            g.create_edge(tids[k], x, y, eid)
            per_type[k][(x, y)].append(eid); eid += 1
        for _ in range(80):
            x, y = int(rng.integers(0, n)), int(rng.integers(0, n))
            if (x, y) in per_type[k]:
                continue
            g.create_edge(tids[k], x, y, eid)
            per_type[k][(x, y)] = [eid]; eid += 1
        for j in rng.choice(len(keys), 120, replace=False):
            x, y = keys[j]
            for e in list(per_type[k].get((x, y), [])):
                g.delete_edge(tids[k], x, y, e)
            per_type[k].pop((x, y), None)
    # mirror the host graph's LAYERS into the oracle graph: for fused chains the reference's delta_lmxm has a
    # row-level mask quirk, so dirty and clean layers give different hop >= 2 results and the
    # oracle must see exactly the (m, dp, dm) the engine sees
    for k in (0, 1):
        me = {p: sorted(es) for p, es in per_type[k].items() if len(es) > 1}
        og.tensors[k] = model.Tensor(n, n, m={(r, c): v for r, c, v in g.layer(tids[k], "m")},
                                     dp={(r, c): v for r, c, v in g.layer(tids[k], "dp")},
                                     dm={(r, c) for r, c, _ in g.layer(tids[k], "dm")}, me=me)
        for p, es in per_type[k].items():
            assert og.tensors[k].get(*p) == sorted(es)
    og.adjacency.m = {(r, c) for r, c, _ in g.layer(None, "m")}
    og.adjacency.dp.layer = {(r, c): True for r, c, _ in g.layer(None, "dp")}
    og.adjacency.dm.layer = {(r, c): True for r, c, _ in g.layer(None, "dm")}
    assert og.adjacency.extract() == set(per_type[0]) | set(per_type[1])
    assert len(og.adjacency.dp.layer) > 0 and len(og.adjacency.dm.layer) > 0      # the layers really are dirty
    assert len(og.tensors[0].dp) > 0 and len(og.tensors[0].dm) > 0
    return g, og, n, per_type


@pytest.fixture(scope="module")
def rnd_graph(hctx):
    return build_random(hctx)


CASES = [
    dict(types=["A"], src_labels=[], dst_labels=[], chain=[]),
    dict(types=[], src_labels=["P"], dst_labels=["Q"], chain=[]),
    dict(types=["A", "B"], src_labels=[], dst_labels=["P"], chain=[]),
    dict(types=["A"], src_labels=["P"], dst_labels=[], chain=[(["B"], ["Q"])]),
    dict(types=[], src_labels=[], dst_labels=[], chain=[([], []), (["A"], ["P"])]),
    dict(types=["B"], src_labels=[], dst_labels=[], chain=[], bind=True),
    dict(types=[], src_labels=["Q"], dst_labels=["P"], chain=[], bind=True),
    dict(types=["A"], src_labels=[], dst_labels=[], chain=[], optional=True),
    dict(types=["Nope"], src_labels=[], dst_labels=[], chain=[], optional=True),
    dict(types=["A"], src_labels=["NoSuchLabel"], dst_labels=[], chain=[]),
    dict(types=["A", "Nope"], src_labels=[], dst_labels=[], chain=[]),                 # [:A|Nope] keeps the A edges (filter_map)
    dict(types=["Nope", "Nada"], src_labels=[], dst_labels=[], chain=[], optional=True),   # no known type: no_match
    dict(types=["A"], src_labels=[], dst_labels=[], chain=[(["Nope", "B"], [])]),      # the same in a fused hop
    dict(types=["B", "Nope"], src_labels=[], dst_labels=[], chain=[], bind=True),      # representative edge scan skips unknowns
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: json.dumps(c, separators=(",", ":"))[:60])
def test_expand_batch_matches_the_oracle(rnd_graph, case):   # cond_traverse.rs:452-751
    g, og, n, _ = rnd_graph
    rng = np.random.default_rng(17)
    src = rng.integers(0, n, 200).tolist()
    hops = [(case["types"], case["dst_labels"])] + [tuple(h) for h in case["chain"]]
    spec = host.cond_spec(src_labels=case["src_labels"], hops=hops, optional=case.get("optional", False),
                          bind=case.get("bind", False))
    got = g.cond_traverse_batch(spec, src)
    want = model.expand_batch(og, src, case["types"], src_labels=case["src_labels"], dst_labels=case["dst_labels"],
                              chain=case["chain"], optional=case.get("optional", False),
                              bind_relationship=case.get("bind", False))
    assert got is not None
    rows, nulls, flops = got
    assert rows == want[0]                                   # emission order: ascending (row, dest)
    assert nulls == want[1]


T_CASES = [
    dict(types=["B"], from_labels=[], to_labels=[]),
    dict(types=["A"], from_labels=["Q"], to_labels=["P"]),
    dict(types=[], from_labels=[], to_labels=["P"]),                  # adjacency: its layers transposed on the device
    dict(types=[], from_labels=["P"], to_labels=[]),
    dict(types=["A", "B"], from_labels=[], to_labels=["Q"]),          # alternation: the materialized union, transposed
    dict(types=["A", "Nope"], from_labels=[], to_labels=[]),
    dict(types=["Nope"], from_labels=[], to_labels=[]),               # unknown type: no rows
]


@pytest.mark.parametrize("case", T_CASES, ids=lambda c: json.dumps(c, separators=(",", ":"))[:60])
def test_expand_batch_over_the_transposed_layers_matches_the_per_row_path(rnd_graph, case):
    """A `transposed` CondTraverse (the bound alias is the matrix destination; the reference serves it row by row over
    build_transposed_iter, cond_traverse.rs:221-235, 840-870) through expand_batch over the transposed device layers — the
    batch against the per-row path of the same library and against oracle/model.py expand_row, emission order and
    representative edge included; then with a pre-bound far end on some rows."""
    g, og, n, per_type = rnd_graph
    rng = np.random.default_rng(23)
    bound = rng.integers(0, n, 150).tolist() + [d for _, d in list(per_type[0])[:50]]
    kw = dict(src_labels=case["from_labels"], hops=[(case["types"], case["to_labels"])])
    spec_rows = host.cond_spec(emit=False, **kw)
    want = []
    for i, b in enumerate(bound):
        want += [(i,) + x for x in model.expand_row(og, b, None, case["types"], from_labels=case["from_labels"],
                                                    to_labels=case["to_labels"], transposed=True, emit_relationship=False)]
    per_row = g.cond_traverse_rows(spec_rows, bound, [None] * len(bound), transposed=True)
    assert per_row == want
    got = g.cond_traverse_batch(host.cond_spec(bind=True, transposed=True, **kw), bound)
    assert got is not None
    rows, nulls, _ = got
    assert [(r, bound[r], d, e) for r, d, e in rows] == want
    assert nulls == []
    # no edge alias: the pairs alone
    rows2, _, _ = g.cond_traverse_batch(host.cond_spec(transposed=True, **kw), bound)
    assert rows2 == [(w[0], w[2]) for w in want]
    if want:
        assert len({w[0] for w in want}) > 20
        # a pre-bound far end on every other row (the reached node must be that one)
        tb = [None] * len(bound)
        first = {}
        for w in want:
            first.setdefault(w[0], w[2])
        for r, d in first.items():
            if r % 2 == 0:
                tb[r] = d
        rows3, _, _ = g.cond_traverse_batch(host.cond_spec(transposed=True, **kw), bound, to_bound=tb)
        assert rows3 == [(w[0], w[2]) for w in want if tb[w[0]] is None or tb[w[0]] == w[2]]
    # OPTIONAL: rows that reach nothing are null-padded
    rows4, nulls4, _ = g.cond_traverse_batch(host.cond_spec(transposed=True, optional=True, **kw), bound)
    assert rows4 == [(w[0], w[2]) for w in want]
    assert nulls4 == sorted(set(range(len(bound))) - {w[0] for w in want})
    # a fused chain is never transposed: the batch declines
    assert g.cond_traverse_batch(host.cond_spec(src_labels=[], hops=[(["A"], []), (["B"], [])], transposed=True), bound[:4]) is None


def test_expand_batch_over_two_coalesced_child_batches(rnd_graph):
    """A host layer that hands the engine TWO child batches at once (2 x 1024 rows; the reference's operator calls down once
    per 1024-row batch, batch.rs:81, cond_traverse.rs:452) must emit, row for row and in the reference's order
    (cond_traverse.rs:644: ascending (row_i, dest)), exactly the two separate calls' rows — the second batch's row indices
    shifted by 1024 — and the oracle's expand_batch over all 2048 rows; 3 x 1024 + 500 rows likewise (the call is then a
    whole frontier for the count form and a plain wide batch for the emitting form)."""
    g, og, n, _ = rnd_graph
    rng = np.random.default_rng(41)
    for hops, kw in (([([], []), (["A"], ["P"])], dict(types=[], chain=[(["A"], ["P"])])),
                     ([(["A"], [])], dict(types=["A"], chain=[])),
                     ([(["B"], []), ([], []), (["A"], [])], dict(types=["B"], chain=[([], []), (["A"], [])]))):
        for total in (2048, 3 * 1024 + 500):
            src = rng.integers(0, n, total).tolist()
            spec = host.cond_spec(src_labels=["P"] if len(hops) == 2 else [], hops=hops)
            both, nulls, flops = g.cond_traverse_batch(spec, src)
            want = []
            fl = 0
            for j in range(0, total, 1024):
                part, _, f = g.cond_traverse_batch(spec, src[j:j + 1024])
                want += [(r + j, d) for r, d in part]
                fl += f
            assert both == want and flops == fl
            ref = model.expand_batch(og, src, kw["types"], src_labels=["P"] if len(hops) == 2 else [], dst_labels=hops[0][1] if not kw["chain"] else [],
                                     chain=kw["chain"])
            assert both == ref[0] and nulls == ref[1]
            assert len(both) > total


def test_expand_batch_to_bound_and_null_sources(rnd_graph):  # cond_traverse.rs:566-575, 657-661
    g, og, n, _ = rnd_graph
    src = [5, None, 9, 300, 17]
    spec = host.cond_spec(hops=[(["A"], [])])
    assert g.cond_traverse_batch(spec, src) is None          # non-node source, not optional: per-row fallback
    assert model.expand_batch(og, src, ["A"]) is None
    spec_o = host.cond_spec(hops=[(["A"], [])], optional=True)
    rows, nulls, _ = g.cond_traverse_batch(spec_o, src)
    want = model.expand_batch(og, src, ["A"], optional=True)
    assert (rows, nulls) == want
    full, _, _ = g.cond_traverse_batch(spec, [5, 9, 300, 17])
    tb = [None, full[0][1] if full else 0, None, 1]
    rows, _, _ = g.cond_traverse_batch(spec, [5, 9, 300, 17], to_bound=tb)
    assert rows == model.expand_batch(og, [5, 9, 300, 17], ["A"], to_bound=tb)[0]
    # every row bound (the multi-hop ExpandInto shape): the operator probes the chain's state instead of expanding the last hop
    spec2 = host.cond_spec(hops=[(["A"], []), ([], [])])
    srcs = [5, 9, 300, 17, 40, 41, 42, 77]
    free, _, _ = g.cond_traverse_batch(spec2, srcs)
    by_row = {}
    for row in free:
        by_row.setdefault(row[0], []).append(row[1])
    tb2 = [by_row[i][len(by_row[i]) // 2] if i in by_row and i % 2 == 0 else (i * 37) % n for i in range(len(srcs))]
    rows, _, _ = g.cond_traverse_batch(spec2, srcs, to_bound=tb2)
    assert rows == model.expand_batch(og, srcs, ["A"], chain=[([], [])], to_bound=tb2)[0]
    assert any(row[0] % 2 == 0 for row in rows)


def test_expand_row_fallback_matches_brute_force(rnd_graph):   # cond_traverse.rs:758-1117
    g, og, n, per_type = rnd_graph
    a = per_type[0]
    srcs = sorted({s for s, _ in a})[:10]
    spec = host.cond_spec(hops=[(["A"], [])], emit=True)
    for s in srcs:
        got = g.cond_traverse_row(spec, from_id=s)
        want = sorted((s, d, e) for (x, d), es in a.items() if x == s for e in es)
        assert sorted(got) == want
    # only the destination bound: walked over the transposed structure
    dsts = sorted({d for _, d in a})[:10]
    for d in dsts:
        got = g.cond_traverse_row(spec, to_id=d)
        want = sorted((x, d, e) for (x, y), es in a.items() if y == d for e in es)
        assert sorted(got) == want
    # anonymous edge: one representative (the smallest id of the first type that has one) per pair
    anon = host.cond_spec(hops=[(["A"], [])], emit=False)
    for s in srcs[:4]:
        got = g.cond_traverse_row(anon, from_id=s)
        want = sorted((s, d, es[0]) for (x, d), es in a.items() if x == s)
        assert sorted(got) == want


ROW_CASES = [
    dict(types=["A"], emit=True),
    dict(types=["A"], emit=False, bidir=True),
    dict(types=[], emit=False, bidir=True, from_labels=["P"], to_labels=["Q"]),
    dict(types=["A", "B"], emit=True, bidir=True, to_labels=["P"]),
    dict(types=["A", "Nope"], emit=False),                       # an alternation drops its unknown names
    dict(types=["Nope"], emit=True),                             # a single unknown type: no rows
    dict(types=["Nope", "Nada"], emit=True, bidir=True),         # an alternation with no known type: no rows
    dict(types=["B"], emit=True, transposed=True, from_labels=["Q"]),
    dict(types=[], emit=False, transposed=True, bidir=True, to_labels=["P"]),
    dict(types=["A"], emit=True, from_labels=["NoSuchLabel"]),
]


@pytest.mark.parametrize("case", ROW_CASES, ids=lambda c: json.dumps(c, separators=(",", ":"))[:70])
def test_expand_row_matches_the_oracle_model(rnd_graph, case):   # cond_traverse.rs:758-1117 vs oracle/model.py expand_row
    """The per-row fallback against the oracle's restatement, emission order included: from-bound, to-bound (walks the
    transposed structure), both bound, neither bound; bidirectional reverse pass; sibling-edge uniqueness; an endpoint
    bound to a non-node."""
    g, og, n, per_type = rnd_graph
    rng = np.random.default_rng(91)
    spec = host.cond_spec(src_labels=case.get("from_labels", []), hops=[(case["types"], case.get("to_labels", []))],
                          emit=case["emit"], bidir=case.get("bidir", False))
    pairs = list(per_type[0])[:40] + list(per_type[1])[:40]
    used = [per_type[0][p][0] for p in list(per_type[0])[:25]]    # ids "already bound to a sibling edge alias"
    rows = [(s, None) for s, _ in pairs[:30]] + [(None, d) for _, d in pairs[30:60]] + pairs[60:75] + \
           [(d, s) for s, d in pairs[:10]] + [(int(rng.integers(0, n)), None) for _ in range(10)] + [("x", None), (3, "x")]
    kw = dict(from_labels=case.get("from_labels", ()), to_labels=case.get("to_labels", ()),
              transposed=case.get("transposed", False), bidirectional=case.get("bidir", False),
              emit_relationship=case["emit"])
    for used_edges in ((), used):
        got = g.cond_traverse_rows(spec, [r[0] for r in rows], [r[1] for r in rows], transposed=case.get("transposed", False),
                                   used_edges=used_edges)
        want = []
        for i, (f, t) in enumerate(rows):
            if isinstance(f, str) or isinstance(t, str):
                continue
            want += [(i,) + x for x in model.expand_row(og, f, t, case["types"], used_edges=used_edges, **kw)]
        assert got == want
    # the whole pair matrix (nothing bound) on the first case only: one long row
    if case is ROW_CASES[1]:
        got = g.cond_traverse_rows(spec, [None], [None])
        assert got == [(0,) + x for x in model.expand_row(og, None, None, case["types"], **kw)]
        assert len(got) > 500


def test_expand_row_bidirectional_dedup_across_rows(rnd_graph):   # cond_traverse.rs:262-299, 948-970
    """An anonymous bidirectional CT whose child is one too keeps ONE row per (scan source, final dest) across the rows of
    an input batch (swap_remove order), with a fresh set for every batch."""
    g, og, n, per_type = rnd_graph
    spec = host.cond_spec(hops=[([], [])], emit=False, bidir=True)
    mids = [s for s, _ in list(per_type[0])[:60]]                  # the intermediate nodes this CT expands from
    scan = [m % 7 for m in mids]                                   # few distinct scan sources: many collisions
    got = g.cond_traverse_rows(spec, mids, [None] * len(mids), dedup_src=scan)
    seen, want = set(), []
    for i, (m, sc) in enumerate(zip(mids, scan)):
        want += [(i,) + x for x in model.expand_row(og, m, None, [], bidirectional=True, dedup=seen, dedup_src=sc)]
    assert got == want
    keys = [(scan[r], t) for r, _, t, _ in got]
    assert len(keys) == len(set(keys))                             # one row per (scan source, dest)
    plain = g.cond_traverse_rows(spec, mids, [None] * len(mids))
    assert len(plain) > len(got) and {(scan[r], t) for r, _, t, _ in plain} == set(keys)
    assert g.cond_traverse_rows(spec, mids, [None] * len(mids), dedup_src=scan) == got   # a new batch starts clean


def test_expand_into_batch_matches_the_oracle(rnd_graph):    # expand_into.rs:121-258
    g, og, n, per_type = rnd_graph
    rng = np.random.default_rng(23)
    pairs = list(per_type[0])[:150] + list(per_type[1])[:150] + \
        [(int(x), int(y)) for x, y in rng.integers(0, n, (100, 2))]
    srcs, dsts = [p[0] for p in pairs], [p[1] for p in pairs]
    for types, bidir, emit in [(["A"], False, True), (["A", "B"], True, True), ([], True, False), (["B"], False, False)]:
        got = g.expand_into(types, srcs, dsts, bidirectional=bidir, emit_relationship=emit, batched=True)
        want = []
        for i, (s, d) in enumerate(pairs):
            want += [(i,) + t for t in model.expand_into_row(og, s, d, types, bidirectional=bidir,
                                                             emit_relationship=emit)]
        assert got == want
        sample = list(range(0, len(pairs), 37))
        one = g.expand_into(types, [srcs[i] for i in sample], [dsts[i] for i in sample], bidirectional=bidir,
                            emit_relationship=emit, batched=False)
        want1 = []
        for j, i in enumerate(sample):
            want1 += [(j,) + t for t in model.expand_into_row(og, srcs[i], dsts[i], types, bidirectional=bidir,
                                                              emit_relationship=emit)]
        assert one == want1


def test_algo_bfs_matches_the_oracle_on_rmat(rnd_graph):     # algo_procedures.rs:1021-1160
    g, og, n, per_type = rnd_graph
    deg = {}
    for (s, d) in og.adjacency.extract():
        deg[s] = deg.get(s, 0) + 1
    src = max(deg, key=deg.get)
    for rel, depth in [(None, -1), ("A", -1), ("B", 2), (None, 1)]:
        got = g.algo_bfs(src, depth, rel, want_edges=True)
        want = model.algo_bfs(og, src, depth, rel, want_edges=True)
        assert (got is None) == (want is None)
        if got is None:
            continue
        assert got[0] == want[0]                             # same reachable set, ascending
        assert len(got[1]) == len(want[1])
        # parents are ANY-valid: every edge returned must end at its node and start one level up
        types = [rel] if rel else []
        adj = og.build_adjacency_matrix(types)
        level, _, _ = oracle.bfs(adj, src, depth, want_parent=False)
        by_edge = {}
        for k in ((0, 1) if rel is None else (og.type_ids[rel],)):
            for (s, d), es in per_type[k].items():
                for e in es:
                    by_edge[e] = (s, d)
        for v, e in zip(got[0], got[1]):
            s, d = by_edge[e]
            assert d == v and level[s] + 1 == level[v]


def test_algo_bfs_from_several_threads_on_one_graph(rnd_graph):
    """the reference's worker pool runs algo.BFS on a shared const Graph from several threads
    (threadpool.rs:89-128).  The cached plan and its pinned level / parent blocks belong to ONE search at a time: a caller
    that finds them busy takes the one-shot path with arrays of its own.  Eight threads x different sources, every result
    must equal the same call made alone."""
    import threading
    g, og, n, per_type = rnd_graph
    deg = {}
    for (s, d) in og.adjacency.extract():
        deg[s] = deg.get(s, 0) + 1
    srcs = sorted(deg, key=deg.get, reverse=True)[:8]
    alone = {s: g.algo_bfs(s, -1, None, want_edges=False) for s in srcs}
    for s in srcs:
        want = model.algo_bfs(og, s, -1, None, want_edges=False)
        assert alone[s][0] == want[0]
    got, errs = {}, []
    gate = threading.Barrier(len(srcs))

    def work(s):
        try:
            gate.wait()
            for _ in range(6):
                r = g.algo_bfs(s, -1, None, want_edges=False)
                if r != alone[s]:
                    errs.append((s, "differs from the call made alone"))
            got[s] = r
        except Exception as e:   # noqa: BLE001
            errs.append((s, repr(e)))
    ts = [threading.Thread(target=work, args=(s,)) for s in srcs]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs[:3]
    assert got == alone


def test_algo_bfs_partitioned_over_a_gang_of_contexts(rnd_graph, hctx):
    """libfalkor_host's algo.BFS over several contexts (here three contexts on the one device): nnz-balanced
    column slabs, level loop + frontier exchange inside libfgpu.so (fgpu_bfs_dist_run) — everything through
    include/fgpu.h — must return what the single-device procedure returns: the same node list, and edges that are valid
    BFS-tree edges."""
    g, og, n, per_type = rnd_graph
    others = [host.Context(0), host.Context(0)]
    gang = [hctx] + others
    try:
        deg = {}
        for (s, d) in og.adjacency.extract():
            deg[s] = deg.get(s, 0) + 1
        for src in (max(deg, key=deg.get), min(deg)):
            for rel, depth in [(None, -1), ("A", -1), ("B", 2)]:
                one = g.algo_bfs(src, depth, rel, want_edges=True)
                got = host.algo_bfs_multi(g, gang, src, depth, rel, want_edges=True)
                assert (got is None) == (one is None)
                if got is None:
                    continue
                assert got[0] == one[0] and len(got[1]) == len(one[1])
                adj = og.build_adjacency_matrix([rel] if rel else [])
                level, _, _ = oracle.bfs(adj, src, depth, want_parent=False)
                by_edge = {}
                for k in ((0, 1) if rel is None else (og.type_ids[rel],)):
                    for (s, d), es in per_type[k].items():
                        for e in es:
                            by_edge[e] = (s, d)
                for v, e in zip(got[0], got[1]):
                    s, d = by_edge[e]
                    assert d == v and level[s] + 1 == level[v]
        assert host.algo_bfs_multi(g, gang, None) is None        # NULL source: no row
    finally:
        for c in others:
            c.close()


# ---- algo.pageRank (algo_procedures.rs:687-783; tests/flow/test_pagerank.py) -----------------------------------
def _model_edge(og, type_id, s, d, eid):
    """the oracle graph is a read-side model: install a committed single edge directly"""
    og.tensors[type_id].m[(s, d)] = eid
    og.adjacency.m.add((s, d))


def _pr_close(got_scores, want_scores):
    # FP32 vectors, FP64-accumulated sums on both sides: 1e-6 relative per score (tests/test_gpu_pagerank.py states the
    # tolerance); only a stopping test that flips one iteration apart may move the scores by ~tol in L1
    g, w = np.asarray(got_scores, dtype=np.float64), np.asarray(want_scores, dtype=np.float64)
    assert g.shape == w.shape
    assert np.allclose(g, w, rtol=1e-6, atol=0) or np.abs(g - w).sum() <= 2e-4


def test_algo_pagerank_reference_flow_cases(hctx):
    # test_pagerank_null_arguments (:40-105): six Node nodes A->B->C->F->E->D->A plus E->B
    names = "ABCDEF"
    idx = {c: i for i, c in enumerate(names)}
    edges = [("A", "B"), ("B", "C"), ("C", "F"), ("F", "E"), ("E", "D"), ("D", "A"), ("E", "B")]
    g, og = host.Graph(hctx, 6), model.Graph(6)
    t, ot = g.add_type("CONNECTS"), og.add_type("CONNECTS")
    l, ol = g.add_label("Node"), og.add_label("Node")
    for v in range(6):
        g.label_node(v, l)
        og.node_labels.add((v, ol))
    for eid, (x, y) in enumerate(edges):
        g.create_edge(t, idx[x], idx[y], eid)
        _model_edge(og, ot, idx[x], idx[y], eid)
    nodes, scores = g.algo_pagerank(None, None)
    wn, ws = model.algo_pagerank(og, None, None)
    assert nodes.tolist() == wn == list(range(6))
    _pr_close(scores, ws)
    assert (scores > 0).all() and abs(float(scores.sum()) - 1.0) < 1e-4
    assert all(scores[idx["B"]] >= s for s in scores)
    # a label that covers every node is the unfiltered run (:711-713)
    n2, s2 = g.algo_pagerank("Node", "CONNECTS")
    assert n2.tolist() == nodes.tolist()
    _pr_close(s2, scores)
    # unknown label / unknown type
    assert len(g.algo_pagerank("Nope", None)[0]) == 0
    n3, s3 = g.algo_pagerank(None, "NoSuchType")
    assert n3.tolist() == list(range(6)) and np.allclose(s3, 1 / 6, rtol=1e-5)     # edgeless: uniform
    # a deleted node stays in the matrix as an isolated vertex and leaves the output (:718-720, :768-770)
    g.delete_node(idx["C"]); og.deleted_nodes.add(idx["C"])
    n4, s4 = g.algo_pagerank(None, None)
    w4n, w4s = model.algo_pagerank(og, None, None)
    assert n4.tolist() == w4n and idx["C"] not in n4.tolist()
    _pr_close(s4, w4s)


def test_algo_pagerank_specific_labels(hctx):
    # test_pagerank_specific_labels (:107-151): A->B (Node, CONNECTS) and S1->S2 (Special, SPECIAL_CONNECTS)
    g, og = host.Graph(hctx, 4), model.Graph(4)
    ln, ls = g.add_label("Node"), g.add_label("Special")
    oln, ols = og.add_label("Node"), og.add_label("Special")
    for v, (lab, olab) in enumerate([(ln, oln), (ln, oln), (ls, ols), (ls, ols)]):
        g.label_node(v, lab)
        og.node_labels.add((v, olab))
    g.create_edge(g.add_type("CONNECTS"), 0, 1, 0); _model_edge(og, og.add_type("CONNECTS"), 0, 1, 0)
    g.create_edge(g.add_type("SPECIAL_CONNECTS"), 2, 3, 1); _model_edge(og, og.add_type("SPECIAL_CONNECTS"), 2, 3, 1)
    nodes, scores = g.algo_pagerank("Special", "SPECIAL_CONNECTS")
    wn, ws = model.algo_pagerank(og, "Special", "SPECIAL_CONNECTS")
    assert nodes.tolist() == wn == [2, 3]
    _pr_close(scores, ws)
    assert scores[1] > scores[0] > 0                         # S2 has the in-edge
    assert len(host.Graph(hctx, 0 + 1).algo_pagerank(None, None)[0]) == 1   # one isolated node: score 1
    empty = host.Graph(hctx, 3)
    for v in range(3):
        empty.delete_node(v)
    assert len(empty.algo_pagerank(None, None)[0]) == 0      # no live node: empty result (:153-160)


def test_algo_pagerank_matches_the_oracle_on_rmat(rnd_graph):
    g, og, n, _ = rnd_graph
    for label, rel in [(None, None), (None, "A"), ("P", None), ("Q", "B")]:
        nodes, scores = g.algo_pagerank(label, rel)
        wn, ws = model.algo_pagerank(og, label, rel)
        assert nodes.tolist() == wn
        _pr_close(scores, ws)
        if len(wn):
            assert abs(float(scores.sum()) - 1.0) < 1e-3


def test_matrix_cursor_new_seek_next(hctx):
    """matrix::Iter (matrix.rs:1471-1605) as a streaming cursor: entries of rows [min, max] ascending, `seek` re-aims the
    same cursor, UINT64 values ride along, a wide sparse range is crossed without per-entry calls, pending writes are
    waited first."""
    rng = np.random.default_rng(4)
    n = 300_000
    rows = np.sort(rng.choice(n, 5000, replace=False)).astype(np.uint64)
    deg = rng.integers(1, 6, len(rows))
    r = np.repeat(rows, deg)
    c = rng.integers(0, n, len(r)).astype(np.uint64)
    v = rng.integers(0, 1 << 60, len(r)).astype(np.uint64)
    m = host.Matrix(hctx, host.Matrix.UINT64, n, n)
    m.build(r, c, v)
    want = {}
    for a, b, x in zip(r.tolist(), c.tolist(), v.tolist()):
        want.setdefault((a, b), x)                           # build keeps the first duplicate? compare through iter below
    eager = m.iter()
    assert [e[:2] for e in eager] == sorted(set(zip(r.tolist(), c.tolist())))
    cur = m.cursor()
    assert list(cur) == eager
    lo, hi = int(rows[100]), int(rows[2100])
    cur.seek(lo, hi)
    assert list(cur) == [e for e in eager if lo <= e[0] <= hi]
    cur.seek(int(rows[7]), int(rows[7]))                     # expand_row's per-source seek (cond_traverse.rs:758-974)
    assert list(cur) == [e for e in eager if e[0] == int(rows[7])]
    cur.seek(n - 1, 2**64 - 1)
    assert list(cur) == [e for e in eager if e[0] == n - 1]
    cur.seek(5, 4)                                           # empty range
    assert list(cur) == []
    m.set(3, 9, 77)                                          # a pending write is visible to a fresh seek (wait first)
    cur.seek(0, 10)
    assert (3, 9, 77) in list(cur)
    b = host.Matrix(hctx, host.Matrix.BOOL, 100, 100)
    b.build(np.array([1, 1, 50], dtype=np.uint64), np.array([2, 3, 99], dtype=np.uint64))
    assert list(b.cursor(1, 50)) == [(1, 2, 1), (1, 3, 1), (50, 99, 1)]


def test_adjacency_builders_match_the_oracle(rnd_graph):    # graph.rs:3870-3907
    g, og, n, _ = rnd_graph
    for types in ([], ["A"], ["A", "B"], ["Nope"]):
        for sym in (False, True):
            got = g.build_adjacency(types, symmetric=sym)
            ref = og.build_symmetric_adjacency_matrix(types) if sym else og.build_adjacency_matrix(types)
            rr, rc = ref.pairs()
            assert [(r, c) for r, c, _ in got.iter()] == list(zip(rr.tolist(), rc.tolist())), (types, sym)


def test_matrix_v19_payload_decode_and_encode(hctx):
    """Decode<19> / Encode<19> for Matrix<T> (matrix.rs:428-546): container payloads built by hand from the documented
    layout decode to the right matrices (sparse / hypersparse, 32- and 64-bit indices, iso BOOL and UINT64 values),
    and what encode() writes parses back to the same arrays and decodes to an equal matrix."""
    from test_host_cpu import _container
    for bits in (32, 64):
        m, used = host.matrix_decode(hctx, _container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6], idx_bits=bits))
        assert (*m.dims(), m.nvals()) == (5, 7, 6)
        assert m.iter() == [(0, 1, 1), (0, 4, 1), (2, 0, 1), (4, 2, 1), (4, 3, 1), (4, 6, 1)]
        big = 3_000_000
        hm, _ = host.matrix_decode(hctx, _container(big, big, [0, 1, 3], [5, 0, big - 1], x=[11, 22, 2 ** 63 + 5],
                                                    h=[3, big - 1], idx_bits=bits))
        assert hm.iter() == [(3, 5, 11), (big - 1, 0, 22), (big - 1, big - 1, 2 ** 63 + 5)]
        assert hm.get(big - 1, 0) == 22 and hm.get(4, 5) is None
    rng = np.random.default_rng(12)
    for typ, n, k in [(host.Matrix.BOOL, 300, 2000), (host.Matrix.UINT64, 300, 2000), (host.Matrix.UINT64, 2_000_000, 500)]:
        r = rng.integers(0, n, k).astype(np.uint64)
        c = rng.integers(0, n, k).astype(np.uint64)
        m = host.Matrix(hctx, typ, n, n)
        m.build(r, c, rng.integers(1, 1 << 62, k).astype(np.uint64) if typ == host.Matrix.UINT64 else None)
        payload = host.matrix_encode(m)
        d = host.container_parse(payload)
        assert d["consumed"] == len(payload) and d["nvals"] == m.nvals() and d["valued"] == (typ == host.Matrix.UINT64)
        assert d["hyper"] == (n == 2_000_000)                       # mostly empty rows are written hypersparse
        back, used = host.matrix_decode(hctx, payload)
        assert used == len(payload) and back.iter() == m.iter()
        assert host.matrix_encode(back) == payload                  # canonical: encode(decode(x)) == x
    e = host.Matrix(hctx, host.Matrix.BOOL, 9, 4)
    back, _ = host.matrix_decode(hctx, host.matrix_encode(e))
    assert (*back.dims(), back.nvals()) == (9, 4, 0)
    # an EMPTY matrix with > 1024 rows is written hypersparse with p = [0], h = [] — what every clean dp / dm layer of
    # a real graph encodes to; the decoder must not mistake the empty hyper list for "not hypersparse"
    for typ in (host.Matrix.BOOL, host.Matrix.UINT64):
        e = host.Matrix(hctx, typ, 2_000_000, 2_000_000)
        payload = host.matrix_encode(e)
        d = host.container_parse(payload)
        assert d["hyper"] and d["nvals"] == 0 and list(d["p"]) == [0] and len(d["h"]) == 0
        back, used = host.matrix_decode(hctx, payload)
        assert used == len(payload) and (*back.dims(), back.nvals()) == (2_000_000, 2_000_000, 0)
        assert back.iter() == [] and back.get(5, 5) is None
        assert host.matrix_encode(back) == payload
    for bits in (32, 64):                                           # the same state built by hand
        hm, _ = host.matrix_decode(hctx, _container(3_000_000, 3_000_000, [0], [], h=[], idx_bits=bits))
        assert (*hm.dims(), hm.nvals()) == (3_000_000, 3_000_000, 0) and hm.iter() == []


# ---- CondVarLenTraverse (cond_var_len_traverse.rs:81-387): trail DFS, device reach sets as pruning oracle ------------
def _twin_graphs(hctx, n, typed_edges, labels=(), commit=True):
    """The same multigraph in the C++ host layer (on the GPU) and in the oracle model: typed_edges = [(type, src, dst)],
    edge ids in list order; labels = [(label, node)]."""
    g, og = host.Graph(hctx, n), model.Graph(n)
    for name, v in labels:
        lid = g.add_label(name); og.add_label(name)
        g.label_node(v, lid); og.node_labels.add((v, og.label_ids[name]))
    pairs = {}
    for eid, (t, s_, d_) in enumerate(typed_edges):
        tid = g.add_type(t); og.add_type(t)
        g.create_edge(tid, s_, d_, eid)
        pairs.setdefault((og.type_ids[t], s_, d_), []).append(eid)
        og.adjacency.m.add((s_, d_))
    for (t, s_, d_), es in pairs.items():
        og.tensors[t].m[(s_, d_)] = es[0] if len(es) == 1 else model.MULTI_EDGE
        if len(es) > 1:
            og.tensors[t].me[(s_, d_)] = es
    if commit:
        g.commit()
    return g, og


def test_var_len_reference_flow_fixtures(hctx):
    """Known answers of tests/flow/test_variable_length_traversals.py, through the host layer on the GPU and the model."""
    # :14-39 the chain A->B->C->D; test02 (:50-62): (a)-[*]->(b) and (a)<-[*]-(b) give 6 rows; test06 (:98-104): 12
    chain = [("knows", i, i + 1) for i in range(3)]
    g, og = _twin_graphs(hctx, 4, chain)
    for rev in (False, True):
        rows = [r for v in range(4) for r in g.var_len_traverse(v, reversed=rev)[0]]
        assert len(rows) == 6 and rows == [r for v in range(4) for r in model.var_len_expand(og, v, reversed=rev)]
    both = [r for v in range(4) for r in g.var_len_traverse(v, bidirectional=True)[0]]
    assert len(both) == 12
    # test05 (:92-96): an unknown relationship type matches nothing; test07 (:106-112): ...except at zero length
    assert [r for v in range(4) for r in g.var_len_traverse(v, types=["no_edge"])[0]] == []
    assert len([r for v in range(4) for r in g.var_len_traverse(v, types=["not_knows"], min_hops=0, max_hops=1)[0]]) == 4
    # test11 (:225-259): a->b->c->a, d->d; undirected paths between a and c
    g, og = _twin_graphs(hctx, 4, [("R", 0, 1), ("R", 1, 2), ("R", 2, 0), ("R", 3, 3)])
    lens = lambda rows: sorted((len(p) - 1) // 2 for (_, _, p) in rows)
    assert lens(g.var_len_traverse(0, dest=2, min_hops=2, max_hops=2, bidirectional=True, emit_path=True)[0]) == [2]
    assert lens(g.var_len_traverse(0, dest=2, min_hops=2, bidirectional=True, emit_path=True)[0]) == [2]
    assert lens(g.var_len_traverse(0, dest=2, bidirectional=True, emit_path=True)[0]) == [1, 2]
    assert lens(g.var_len_traverse(3, min_hops=0, max_hops=0, bidirectional=True, emit_path=True)[0]) == [0]
    # test12 (:261-288): a->b->c->a plus a->d; (a)-[*2..]->(z) does not get stuck in the cycle: z = a, c and one more (d)
    g, og = _twin_graphs(hctx, 4, [("R", 0, 1), ("R", 1, 2), ("R", 2, 0), ("R", 0, 3)])
    z = sorted(t for (_, t, _) in g.var_len_traverse(0, min_hops=2)[0])
    assert len(z) == 3 and z[0] == 0 and 2 in z
    # test13 (:290-339): a tree of fanout 3, depth 2: (root)-[*0..]->(n) reaches all 13 nodes once
    tree = [("R", 0, 1 + i) for i in range(3)] + [("R", 1 + i, 4 + 3 * i + j) for i in range(3) for j in range(3)]
    g, og = _twin_graphs(hctx, 13, tree)
    rows, _ = g.var_len_traverse(0, min_hops=0)
    assert sorted(t for (_, t, _) in rows) == list(range(13))
    assert rows == model.var_len_expand(og, 0, min_hops=0)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("commit", [True, False], ids=["committed", "pending-deltas"])
def test_var_len_matches_model_on_random_multigraphs(hctx, seed, commit):
    """Emission order, endpoints and paths against oracle/model.py var_len_expand on small multigraphs with two types,
    multi-edges, self-loops and cycles: every direction, hop ranges, bound destination (pruned and unpruned: identical
    rows, and the reach sets do cut frames), destination labels, an alternation with an unknown type."""
    rng = np.random.default_rng(seed)
    n = 9
    edges = [(("KNOWS", "LIKES")[int(rng.integers(0, 2))], int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(16)]
    edges += [edges[0], edges[3]]                                  # multi-edge pairs
    labels = [("L", v) for v in range(0, n, 2)]
    g, og = _twin_graphs(hctx, n, edges, labels, commit)
    cut = 0
    for start in (0, 3, 7):
        for kw in ({}, {"reversed": True}, {"bidirectional": True}):
            for (lo, hi) in ((1, 1), (1, 3), (2, 4), (0, 2), (1, None)):
                if hi is None and kw.get("bidirectional"):
                    hi = 5                                          # unbounded undirected trails explode on 18 edges
                want = model.var_len_expand(og, start, min_hops=lo, max_hops=hi, emit_path=True, **kw)
                got, _ = g.var_len_traverse(start, min_hops=lo, max_hops=hi, emit_path=True, **kw)
                assert got == want, (start, kw, lo, hi)
                for dest in (2, start):
                    want = model.var_len_expand(og, start, dest=dest, min_hops=lo, max_hops=hi, emit_path=True, **kw)
                    got, st = g.var_len_traverse(start, dest=dest, min_hops=lo, max_hops=hi, emit_path=True, **kw)
                    raw, st0 = g.var_len_traverse(start, dest=dest, min_hops=lo, max_hops=hi, emit_path=True, prune=False, **kw)
                    assert got == want == raw, (start, dest, kw, lo, hi)
                    assert st["frames"] <= st0["frames"] and st0["pruned"] == 0
                    cut += st0["frames"] - st["frames"]
            want = model.var_len_expand(og, start, types=["KNOWS"], max_hops=3, dst_labels=["L"], **kw)
            assert g.var_len_traverse(start, types=["KNOWS"], max_hops=3, dst_labels=["L"], **kw)[0] == want
            want = model.var_len_expand(og, start, types=["LIKES", "NOPE"], max_hops=2, **kw)
            assert g.var_len_traverse(start, types=["LIKES", "NOPE"], max_hops=2, **kw)[0] == want
            assert g.var_len_traverse(start, max_hops=2, dst_labels=["MISSING"], **kw)[0] == []
    assert cut > 0                                                  # the device reach sets did prune something
