"""CPU restatement of LAGr_PageRank (LAGraph v1.3.x, un-vendored: /root/reference links it as a static library,
graph/build.rs:50-52) as algo.pageRank calls it — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Call site: graph/src/runtime/functions/algo_procedures.rs:694-783 — `LAGr_PageRank(&centrality, &iters, G, 0.85,
1e-4, 100, msg)` on `build_adjacency_matrix(rel_types)` resized to node_count + deleted_nodes (unfiltered), or on
the compact adjacency of the labelled nodes (`build_compact_adj_from_tensors`, :725-733); G->AT and
G->out_degree are cached first (:736-739).  The algorithm (LAGraph's documented "standard PageRank, sinks handled
properly", binding doc lagraph_bindings.rs:549), all vectors GrB_FP32:

    r = 1/n ;  d = max(1/damping, out_degree/damping)  ;  sinks = vertices with no out_degree entry
    for iters in 0..itermax while rdiff > tol:
        teleport = (1-damping)/n  [+ (damping/n) * sum(r[sinks]) when there are sinks]
        t <-> r ;  w = t ./ d ;  r = teleport ;  r += A' (plus_second) w
        t -= r ; t = |t| ; rdiff = sum(t)

Arithmetic model: the vectors are FP32 (GrB_FP32), every sum — a row of the plus_second product, the sink mass, rdiff —
is accumulated in FP64 here and rounded to FP32 once.  The reference accumulates in FP32 in a GraphBLAS-internal order,
so each of its sums carries a rounding error of up to ~sqrt(k) ulps that depends on an order nobody specifies; the
FP64-accumulated value is the centre of that cloud and the one number every legal FP32 order approximates.  The engine
(pagerank.hip) accumulates the same way, so engine and oracle agree to FP32 rounding of identical values: the -m gpu test
holds 1e-6 relative per score (north_star's PLUS_TIMES tolerance); against an FP32-order reference the expected
deviation is the reference's own rounding, ~1e-6 on hub rows.

PARITY UNPINNED beyond the properties the reference's flow test holds (tests/flow/test_pagerank.py:40-151: one
score per node, scores sum to 1 +- 1e-4, all positive, the node with two in-edges ranks highest, S2 > S1):
LAGraph's source is absent, so this is a restatement from its published algorithm, and GraphBLAS' summation
order is internal — results are defined up to FP32 rounding.
"""
from __future__ import annotations

import numpy as np

from . import CSR, build_csr, transpose

F32 = np.float32


def _row_sums(at: CSR, w: np.ndarray) -> np.ndarray:
    """sum of w over every row's column ids, FP64 accumulation (returned as float64; the caller rounds once)."""
    n = at.nrows
    out = np.zeros(n, dtype=np.float64)
    if at.nnz == 0:
        return out
    rp = at.rowptr.astype(np.int64)
    vals = w[at.colidx.astype(np.int64)].astype(np.float64)
    nz = np.nonzero(np.diff(rp))[0]
    out[nz] = np.add.reduceat(vals, rp[nz])
    return out


def pagerank(a: CSR, damping=0.85, tol=1e-4, itermax=100, active=None):
    """(scores float32[n], iters).  `active` (bool[n] or None): run on the induced subgraph of the flagged vertices
    (the label-filtered form), 0 elsewhere."""
    n_all = a.nrows
    if active is not None:
        active = np.asarray(active, dtype=bool)
        ids = np.nonzero(active)[0]
        remap = np.full(n_all, -1, dtype=np.int64)
        remap[ids] = np.arange(len(ids))
        rows, cols = a.pairs()
        keep = active[rows.astype(np.int64)] & active[cols.astype(np.int64)]
        sub = build_csr(len(ids), len(ids), remap[rows[keep].astype(np.int64)].astype(np.uint64),
                        remap[cols[keep].astype(np.int64)].astype(np.uint64))
        out = np.zeros(n_all, dtype=F32)
        if len(ids) == 0:
            return out, 0
        s, it = pagerank(sub, damping, tol, itermax)
        out[ids] = s
        return out, it
    n = n_all
    if n == 0:
        return np.zeros(0, dtype=F32), 0
    damping, tol = F32(damping), F32(tol)
    at = transpose(a)
    deg = np.diff(a.rowptr.astype(np.int64)).astype(F32)
    sink = deg == 0
    nsinks = int(sink.sum())
    r = np.full(n, F32(1.0) / F32(n), dtype=F32)
    t = np.zeros(n, dtype=F32)
    d = np.maximum(F32(1.0) / damping, deg / damping).astype(F32)
    teleport0 = (F32(1.0) - damping) / F32(n)
    damp_over_n = damping / F32(n)
    rdiff, iters = F32(1.0), 0
    while iters < itermax and rdiff > tol:
        teleport = teleport0
        if nsinks:
            teleport = F32(np.float64(teleport0) + np.float64(damp_over_n) * r[sink].sum(dtype=np.float64))
        t, r = r, t
        w = (t / d).astype(F32)
        r = (np.float64(teleport) + _row_sums(at, w)).astype(F32)
        rdiff = F32(np.abs(t.astype(np.float64) - r.astype(np.float64)).sum(dtype=np.float64))
        iters += 1
    return r, iters
