"""Pure-Python restatement of the reference's Delta-matrix layer and traversal operators.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): small cases, obvious code, one reference
citation per method (file:line relative to /root/reference).

    VersionedMatrix   graph/src/graph/graphblas/versioned_matrix.rs  ("Delta_Matrix")
    Tensor (read)     graph/src/graph/graphblas/tensor.rs:154-319, 841-943
    Graph             graph/src/graph/graph.rs (traversal-facing parts only)
    expand_batch      graph/src/runtime/ops/cond_traverse.rs:452-751
    expand_into_row   graph/src/runtime/ops/expand_into.rs:121-258
    algo_bfs          graph/src/runtime/functions/algo_procedures.rs:1021-1160
    algo_pagerank     graph/src/runtime/functions/algo_procedures.rs:687-783
    fuse_anonymous_traverse  graph/src/planner/optimizer/fuse_anonymous_traverse.rs:83-284
"""
from __future__ import annotations

import numpy as np

from . import CSR, bfs as _bfs, build_csr, delta_lmxm, empty

# ---- fold policy (versioned_matrix.rs:140-188) -------------------------------------------------
WRITE_FOLD_K = 20_500_000
READ_FOLD_K = 82_000
MIN_FOLD_DELTA = 256
U64_MAX = (1 << 64) - 1


def _sat_mul(a, b):
    return min(a * b, U64_MAX)


def fold_balance(delta_nvals, tx_added, base_nvals, k):
    """versioned_matrix.rs:175-188"""
    return (tx_added > 0 and delta_nvals >= MIN_FOLD_DELTA
            and (_sat_mul(delta_nvals, 2) >= base_nvals or _sat_mul(delta_nvals, delta_nvals) >= _sat_mul(k, tx_added)))


def should_fold(delta_nvals, tx_added, base_nvals):       # versioned_matrix.rs:152-158 (dup / write path)
    return fold_balance(delta_nvals, tx_added, base_nvals, WRITE_FOLD_K)


def should_fold_read(delta_nvals, tx_added, base_nvals):  # versioned_matrix.rs:164-170 (wait / read path)
    return fold_balance(delta_nvals, tx_added, base_nvals, READ_FOLD_K)


def delta_dominates_base(delta_nvals, base_nvals):        # versioned_matrix.rs:195-200
    return delta_nvals >= MIN_FOLD_DELTA and _sat_mul(delta_nvals, 2) >= base_nvals


class Delta:
    """One delta layer + the fold bookkeeping (versioned_matrix.rs:214-478)."""

    def __init__(self, entries=None):
        self.layer = dict(entries or {})   # (i, j) -> value
        self.count = len(self.layer)       # approximate nvals, maintained without materializing
        self.tx_nvals = 0
        self.fold = False
        self.pending = False               # Matrix::has_pending of the layer

    def new_version(self, fold):           # :337-349
        d = Delta(self.layer)
        d.count = self.count
        d.tx_nvals = self.count
        d.fold = fold
        d.pending = self.pending
        return d

    def is_synced(self):
        return not self.pending

    def resync(self):                      # :355-358
        self.pending = False
        self.count = len(self.layer)

    def latch(self, decision):             # :360-367
        if decision:
            self.fold = True

    def fold_decision(self, policy, base):  # :369-377
        return self.fold or policy(self.count, max(self.count - self.tx_nvals, 0), base)

    def take_fold(self):                   # :383-385
        f, self.fold = self.fold, False
        return f and len(self.layer) > 0

    def clear(self):                       # :387-398
        self.layer = {}
        self.count = 0
        self.tx_nvals = 0
        self.fold = False
        self.pending = False

    def insert(self, i, j, value=True):    # :429-437 (count moves whether or not the key existed)
        self.layer[(i, j)] = value
        self.count += 1
        self.pending = True

    def erase(self, i, j):                 # :414-422
        self.layer.pop((i, j), None)
        self.count = max(self.count - 1, 0)
        self.pending = True


class VersionedMatrix:
    """VersionedMatrix<bool>: base m + pending adds dp + tombstones dm (versioned_matrix.rs:480-1079).
    Invariants: dp ∩ m = ∅, dm ⊆ m, dp ∩ dm = ∅."""

    def __init__(self, nrows, ncols, m=None):
        self.nrows, self.ncols = nrows, ncols
        self.m = set(m or ())
        self.dp = Delta()
        self.dm = Delta()
        self.needs_flush = False

    @classmethod
    def from_matrix(cls, nrows, ncols, entries):    # :877-890
        return cls(nrows, ncols, entries)

    def wait(self):                                 # :545-556
        if self.dp.is_synced() and self.dm.is_synced():
            return
        self.dp.resync()
        self.dm.resync()
        base = len(self.m)
        self.dp.latch(self.dp.fold_decision(should_fold_read, base))
        self.dm.latch(self.dm.fold_decision(should_fold_read, base))

    def wait_all(self):                             # :562-566
        self.dp.pending = False
        self.dm.pending = False

    def nvals(self):                                # :629-632
        self.wait()
        return len(self.m) + len(self.dp.layer) - len(self.dm.layer)

    def extract(self):                              # :609-620  pattern(m) \ dm U pattern(dp)
        self.wait()
        return (self.m - set(self.dm.layer)) | set(self.dp.layer)

    def get(self, i, j):                            # :819-835
        self.wait()
        if (i, j) in self.m:
            return None if (i, j) in self.dm.layer else True
        return self.dp.layer.get((i, j))

    def iter(self, min_row=0, max_row=U64_MAX):     # :647-654 + Iter :1116-1253 (3-way sorted merge)
        self.wait()
        return sorted(e for e in self.extract() if min_row <= e[0] <= max_row)

    def flush(self):                                # :892-938
        if not self.needs_flush:
            return
        self.wait_all()
        fold_dp = self.dp.take_fold()
        fold_dm = self.dm.take_fold()
        if fold_dp and fold_dm:
            self.m = (self.m | set(self.dp.layer)) - set(self.dm.layer)   # eWiseAdd<!dm>(m, dp), :905-911
        elif fold_dp:
            self.m = self.m | set(self.dp.layer)
        elif fold_dm:
            self.m = self.m - set(self.dm.layer)                          # select(!dm, m)
        if fold_dp:
            self.dp.clear()
        if fold_dm:
            self.dm.clear()
        self.needs_flush = False

    def set(self, i, j, value=True):                # :844-857
        self.flush()
        if (i, j) in self.m:
            self.dm.erase(i, j)
        else:
            self.dp.insert(i, j)

    def remove(self, i, j):                         # :780-791
        self.flush()
        if (i, j) in self.m:
            self.dm.insert(i, j)
        else:
            self.dp.erase(i, j)

    def remove_mask(self, mask):                    # :799-816   dm U= mask ∩ m ; dp \= mask
        self.flush()
        mask = set(mask)
        for e in mask & self.m:
            self.dm.layer[e] = True
        self.dm.resync()                            # tombstone_masked resyncs (:439-447)
        for e in mask:
            self.dp.layer.pop(e, None)
        self.dp.resync()                            # remove_all resyncs (:451-458)

    def set_all(self, entries, new=False):          # :1006-1035
        self.flush()
        self.dm.pending = False                     # dm.wait()
        if len(self.dm.layer) == 0:
            for (i, j) in entries:
                if not new and (i, j) in self.m:
                    continue
                self.dp.insert(i, j)
        else:
            for (i, j) in entries:
                self.set(i, j, True)

    def dup(self):                                  # :1038-1051
        base = len(self.m)
        fold_dp = self.dp.fold_decision(should_fold, base)
        fold_dm = self.dm.fold_decision(should_fold, base)
        v = VersionedMatrix(self.nrows, self.ncols, self.m)
        v.dp = self.dp.new_version(fold_dp)
        v.dm = self.dm.new_version(fold_dm)
        v.needs_flush = fold_dp or fold_dm
        return v

    def fold_oversized(self):                       # :953-965
        base = len(self.m)
        odp = delta_dominates_base(self.dp.count, base)
        odm = delta_dominates_base(self.dm.count, base)
        if odp or odm:
            self.dp.latch(odp)
            self.dm.latch(odm)
            self.needs_flush = True
            self.flush()

    # layers as CSR for the products
    def layers(self):
        self.wait()
        return (_csr(self.nrows, self.ncols, self.m), _csr(self.nrows, self.ncols, self.dp.layer),
                _csr(self.nrows, self.ncols, self.dm.layer))


def _csr(nrows, ncols, pairs) -> CSR:
    pairs = list(pairs)
    if not pairs:
        return empty(nrows, ncols)
    a = np.asarray(pairs, dtype=np.uint64)
    return build_csr(nrows, ncols, a[:, 0], a[:, 1])


MULTI_EDGE = U64_MAX  # tensor.rs:207


class Tensor:
    """Read side of a per-relationship-type Tensor (tensor.rs:184-319, 841-943): forward u64 layers
    whose value is the inline edge id or MULTI_EDGE, multi-edge ids in `me`, bool transpose `mt`."""

    def __init__(self, nrows, ncols, m=None, dp=None, dm=None, me=None):
        self.nrows, self.ncols = nrows, ncols
        self.m = dict(m or {})       # (s, d) -> id | MULTI_EDGE
        self.dp = dict(dp or {})
        self.dm = set(dm or ())
        self.me = {k: sorted(v) for k, v in (me or {}).items()}  # (s, d) -> ascending edge ids

    def eff_get(self, s, d):         # tensor.rs:286-299
        if (s, d) in self.dp:
            return self.dp[(s, d)]
        if (s, d) in self.dm:
            return None
        return self.m.get((s, d))

    def get(self, s, d):             # tensor.rs:307-319 -> ascending edge ids
        v = self.eff_get(s, d)
        if v is None:
            return []
        if v == MULTI_EDGE:
            return list(self.me.get((s, d), []))
        return [v]

    def structure(self):             # (pattern(m) \ dm) U pattern(dp); mt is its transpose (tensor.rs:886-888)
        return (set(self.m) - self.dm) | set(self.dp)

    def fwd_layers(self):            # fwd_m / fwd_dp / fwd_dm (tensor.rs:841-856) as patterns
        return (_csr(self.nrows, self.ncols, self.m), _csr(self.nrows, self.ncols, self.dp),
                _csr(self.nrows, self.ncols, self.dm))


class TensorPairs:
    """Pair-level model of a Tensor (tensor.rs:184-989): what every layer state (m / dp / dm / me, the diagram at
    :72-108) must look like from outside — (src, dst) -> ascending edge ids.  The reference's own unit tests
    (tensor.rs:1340-1669) are stated on exactly these observables plus layer sizes; the observables are replayed here
    (tests/test_oracle_golden.py) and, with the layer sizes, on the product (tests/test_gpu_host.py)."""
    MULTI_EDGE = U64_MAX

    def __init__(self):
        self.pairs = {}

    def set_all_from_slices(self, srcs, dsts, ids):          # :333-455: duplicates of a pair in one batch promote it
        for s, d, e in zip(srcs, dsts, ids):
            row = self.pairs.setdefault((int(s), int(d)), [])
            if int(e) not in row:
                row.append(int(e))
                row.sort()

    def remove_all(self, rels):                              # :461-657 -> the pairs left without an edge, each once
        emptied = []
        for e, s, d in rels:
            row = self.pairs.get((int(s), int(d)))
            if row is None or int(e) not in row:
                continue                                     # foreign / repeated ids change nothing (:1590-1630)
            row.remove(int(e))
            if not row:
                del self.pairs[(int(s), int(d))]
                emptied.append((int(s), int(d)))
        return emptied

    def get(self, s, d):                                     # :307-319
        return list(self.pairs.get((s, d), []))

    def eff_get(self, s, d):                                 # :286-299: inline id, or the sentinel for a multi pair
        row = self.pairs.get((s, d))
        if not row:
            return None
        return row[0] if len(row) == 1 else self.MULTI_EDGE

    def edge_count(self):                                    # :955-967
        return sum(len(r) for r in self.pairs.values())

    def multi_pairs(self):
        return sum(1 for r in self.pairs.values() if len(r) > 1)

    def me_nvals(self):                                      # ids held in `me`: those of the multi pairs
        return sum(len(r) for r in self.pairs.values() if len(r) > 1)

    def extract(self):                                       # :838-850 pattern
        return set(self.pairs)


class Graph:
    """The traversal-facing slice of graph.rs: adjacency, node-label matrix, per-type tensors."""

    def __init__(self, node_cap):
        self.n = node_cap
        self.adjacency = VersionedMatrix(node_cap, node_cap)    # graph.rs:2251
        self.node_labels = set()                                # (node, label_id)   graph.rs:1057-1066
        self.tensors: list[Tensor] = []                         # relationship_matrices  graph.rs:2256
        self.label_ids: dict[str, int] = {}
        self.type_ids: dict[str, int] = {}
        self.deleted_nodes: set[int] = set()

    def add_label(self, name):
        return self.label_ids.setdefault(name, len(self.label_ids))

    def add_type(self, name):
        if name not in self.type_ids:
            self.type_ids[name] = len(self.tensors)
            self.tensors.append(Tensor(self.n, self.n))
        return self.type_ids[name]

    def node_has_label_id(self, node, lid):     # graph.rs:1057-1066
        return (node, lid) in self.node_labels

    def resolve_label_ids(self, labels):        # graph.rs:2554-2559: unknown label -> None (no rows)
        out = []
        for l in labels:
            if l not in self.label_ids:
                return None
            out.append(self.label_ids[l])
        return out

    def traversal_layers(self, types):
        """Matrix choice of expand_batch (cond_traverse.rs:478-505): [] -> adjacency; one type -> that
        Tensor's forward layers; several -> materialized union with clean deltas
        (build_relationship_matrix_unrestricted, graph.rs:2520-2549).  A single unknown type -> None
        (:481-490); in an alternation the unknown names are filtered out (`filter_map`, graph.rs:2524-2527) and
        only an alternation with NO known type gives None — `[:KNOWS|NOPE]` still returns the KNOWS edges, through
        the materialized union (clean deltas), even when one type is left."""
        if not types:
            return self.adjacency.layers()
        if len(types) == 1:
            if types[0] not in self.type_ids:
                return None
            return self.tensors[self.type_ids[types[0]]].fwd_layers()
        ids = [self.type_ids[t] for t in types if t in self.type_ids]
        if not ids:
            return None
        u = set()
        for i in ids:
            u |= self.tensors[i].structure()
        return _csr(self.n, self.n, u), empty(self.n, self.n), empty(self.n, self.n)

    def build_adjacency_matrix(self, types) -> CSR:   # graph.rs:3870-3894
        if not types:
            return _csr(self.n, self.n, self.adjacency.extract())
        u = set()
        for t in types:
            if t in self.type_ids:
                u |= self.tensors[self.type_ids[t]].structure()
        return _csr(self.n, self.n, u)

    def build_symmetric_adjacency_matrix(self, types) -> CSR:   # graph.rs:3898-3907: A (+) A'
        a = self.build_adjacency_matrix(types)
        r, c = a.pairs()
        return build_csr(self.n, self.n, np.concatenate([r, c]), np.concatenate([c, r]))

    def get_src_dest_relationships(self, s, d, types):   # graph.rs:1797-1837 (ids in type order)
        ids = []
        tids = [self.type_ids[t] for t in types if t in self.type_ids] if types else range(len(self.tensors))
        for t in tids:
            ids.extend(self.tensors[t].get(s, d))
        return ids


def expand_batch(g: Graph, src_values, types, src_labels=(), dst_labels=(), chain=(), optional=False,
                 to_bound=None, bind_relationship=False):
    """CondTraverseOp::expand_batch (cond_traverse.rs:452-751), SURVEY Appendix A.2.

    src_values[i]: node id (int) or None (bound to a non-node, e.g. NULL).  chain: sequence of
    (types, dst_labels) for the fused hops.  Returns (rows, null_rows): rows = list of
    (active_row_index, dest[, edge_id]) in emission order; null_rows = unmatched active rows that an
    optional traverse null-pads (appended after the matches, :737-747).  Returns None when the
    batched path bails to the per-row fallback (non-node source on a non-optional traverse)."""
    k = len(src_values)
    hops = [(types, dst_labels)] + [tuple(h) for h in chain]
    layers = []
    for ht, _ in hops:
        l = g.traversal_layers(list(ht))
        if l is None:                                # unknown type: no_match (:485-490)
            return ([], list(range(k)) if optional else [])
        layers.append(l)
    last_dst = g.resolve_label_ids(list(hops[-1][1]))
    src_lids = g.resolve_label_ids(list(src_labels))
    if last_dst is None or src_lids is None:
        return ([], list(range(k)) if optional else [])
    rows, cols = [], []
    for i, v in enumerate(src_values):
        if v is None:                                # bound to a non-Node (:566-575)
            if optional:
                continue
            return None
        if not all(g.node_has_label_id(v, l) for l in src_lids):   # :580-586
            continue
        rows.append(i)
        cols.append(v)
    if not rows:
        return ([], list(range(k)) if optional else [])
    f = build_csr(k, g.n, rows, cols)                # :600-601
    for (m, dp, dm) in layers:                       # :602-605
        f, _ = delta_lmxm(f, m, dp, dm)
    out = []
    matched = [False] * k
    fr, fc = f.pairs()
    for i, d in zip(fr.tolist(), fc.tolist()):       # ascending (row, col)  (:644)
        if not all(g.node_has_label_id(d, l) for l in last_dst):   # :647-651
            continue
        if to_bound is not None and to_bound[i] is not None and to_bound[i] != d:   # :657-661
            continue
        if bind_relationship and len(hops) == 1:     # :663-695 representative edge, first type with an id
            ids = g.get_src_dest_relationships(src_values[i], d, list(types))
            if not ids:
                continue
            out.append((i, d, ids[0]))
        else:
            out.append((i, d))
        matched[i] = True
    nulls = [i for i in range(k) if not matched[i]] if optional else []
    return out, nulls


def expand_row(g: Graph, from_id, to_id, types, from_labels=(), to_labels=(), transposed=False, bidirectional=False,
               emit_relationship=False, used_edges=(), dedup=None, dedup_src=None):
    """CondTraverseOp::expand_row + process_pairs (cond_traverse.rs:758-974, 978-1117), the per-row fallback, without the
    attribute filters: returns [(from, to, edge id)] in emission order.

    build_state (:362-440): the pair matrix is the adjacency ([] types), one tensor's structure, or the union of the
    KNOWN types of an alternation (None known / a single unknown type -> no rows); an unknown label -> no rows.
    Forward pass: the rows of the matrix source, or — only the matrix destination bound — that node's row of the
    TRANSPOSED matrix ((dest, src) ascending, :221-235, 852-864).  Bidirectional: a second pass with the endpoints
    swapped, self-loops dropped (:894-945).  process_pairs: label checks on the raw matrix coordinates, from / to
    filters on the oriented pair, one representative edge per pair unless emit_relationship, ids already bound to a
    sibling edge alias skipped.  `dedup` (a set shared by the rows of one input batch): rows whose (dedup_src, to) was
    seen are removed with swap_remove (:948-970)."""
    types = list(types)
    if not types:
        struct = set(g.adjacency.extract())
        scan = list(range(len(g.tensors)))
    elif len(types) == 1:
        if types[0] not in g.type_ids:
            return []
        struct = g.tensors[g.type_ids[types[0]]].structure()
        scan = [g.type_ids[types[0]]]
    else:
        scan = [g.type_ids[t] for t in types if t in g.type_ids]
        if not scan:
            return []
        struct = set()
        for t in scan:
            struct |= g.tensors[t].structure()
    from_l, to_l = g.resolve_label_ids(list(from_labels)), g.resolve_label_ids(list(to_labels))
    if from_l is None or to_l is None:
        return []
    fwd = sorted(struct)                                       # ascending (row, col)
    bwd = sorted((d, s) for (s, d) in struct)                  # the transposed matrix, ascending (dest, src)

    def pairs_of(msrc, mdst, drop_loops):
        if msrc is None and mdst is not None:
            ps = [(s, d) for (d, s) in bwd if d == mdst]
        else:
            ps = [(s, d) for (s, d) in fwd if (msrc is None or s == msrc) and (mdst is None or d == mdst)]
        return [(s, d) for (s, d) in ps if not (drop_loops and s == d)]

    out = []

    def process(pairs, is_reverse, sl, dl):
        for (s, d) in pairs:
            if not all(g.node_has_label_id(s, l) for l in sl) or not all(g.node_has_label_id(d, l) for l in dl):
                continue
            fn, tn = (d, s) if is_reverse else (s, d)
            if from_id is not None and from_id != fn:
                continue
            if to_id is not None and to_id != tn:
                continue
            done = False
            for t in scan:
                for e in g.tensors[t].get(s, d):
                    if e in used_edges:
                        continue
                    out.append((fn, tn, e))
                    if not emit_relationship:
                        done = True
                        break
                if done:
                    break

    fwd_src, fwd_dst = (to_id, from_id) if transposed else (from_id, to_id)
    process(pairs_of(fwd_src, fwd_dst, False), transposed, to_l if transposed else from_l, from_l if transposed else to_l)
    if bidirectional:
        rev_src, rev_dst = (from_id, to_id) if transposed else (to_id, from_id)
        process(pairs_of(rev_src, rev_dst, True), not transposed, from_l if transposed else to_l,
                to_l if transposed else from_l)
    if dedup is not None and dedup_src is not None:
        i = 0
        while i < len(out):
            key = (dedup_src, out[i][1])
            if key in dedup:
                out[i] = out[-1]
                out.pop()
                continue
            dedup.add(key)
            i += 1
    return out


def node_relationships(g: Graph, node, types, outgoing=True, incoming=False):
    """Graph::get_node_relationships_by_type (graph.rs:1797-1835): [(src, dst, edge id)] — per tensor (all of them, or
    the KNOWN ones among `types`, in the order given) the outgoing half (dst, id ascending: Tensor::iter(id, id, false)),
    then the incoming half (src, id ascending: the transposed iterator), without the self-loops the outgoing half
    already produced."""
    tids = [g.type_ids[t] for t in types if t in g.type_ids] if types else list(range(len(g.tensors)))
    out = []
    for t in tids:
        T = g.tensors[t]
        struct = sorted(T.structure())
        if outgoing:
            for (s_, d_) in struct:
                if s_ == node:
                    out.extend((s_, d_, e) for e in T.get(s_, d_))
        if incoming:
            for (d_, s_) in sorted((d_, s_) for (s_, d_) in struct):
                if d_ == node and not (outgoing and s_ == node):
                    out.extend((s_, d_, e) for e in T.get(s_, d_))
    return out


def var_len_expand(g: Graph, start, types=(), dest=None, min_hops=1, max_hops=None, reversed=False, bidirectional=False,
                   dst_labels=(), emit_path=False):
    """CondVarLenTraverse over one input row (cond_var_len_traverse.rs:81-387, VarLenIter): [(from, to, path | None)] in
    the reference's emission order.  Trail semantics: a relationship id is used at most once on a path, nodes may repeat
    (:123-126).  Frames leave a LIFO stack (:221); a frame walks its node's cached adjacency (:240-243), keeps the
    neighbours the direction allows (:253-266), and for each, in adjacency order, emits when hop >= min_hops and the
    destination / label filters pass (:316-319), and pushes a continuation when hop < max_hops (:321); the frame's
    emissions are yielded before the next frame runs (:379-383).  A reversed traversal walks incoming edges and reports
    (neighbour, start) (:343-347); its path is reversed into pattern order (:134-142).  Attribute / WHERE edge filters are
    out of scope (no attribute store)."""
    types = list(types)
    out = []
    lids = []
    label_missing = False
    for l in dst_labels:
        if l in g.label_ids:
            lids.append(g.label_ids[l])
        else:
            label_missing = True

    def labels_ok(v):
        return all(g.node_has_label_id(v, l) for l in lids)

    def emit(other, walk):
        path = None
        if emit_path:
            path = list(walk[::-1]) if reversed else list(walk)
        out.append((other, start, path) if reversed else (start, other, path))

    if min_hops == 0 and (dest is None or dest == start) and not label_missing and labels_ok(start):   # :153-171
        emit(start, [start])
    w_out, w_in = bidirectional or not reversed, bidirectional or reversed
    cache = {}
    stack = [(start, [start] if emit_path else [], (), 0)]
    while stack:
        node, walk, used, depth = stack.pop()
        hop = depth + 1
        if max_hops is not None and hop > max_hops:
            continue
        if node not in cache:
            cache[node] = node_relationships(g, node, types, w_out, w_in)
        nbrs = []
        for (s_, d_, e) in cache[node]:
            if e in used:
                continue
            if reversed:
                if d_ == node:
                    nbrs.append((e, s_))
            elif s_ == node:
                nbrs.append((e, d_))
            elif bidirectional and d_ == node:
                nbrs.append((e, s_))
        for (e, nb) in nbrs:
            will_emit = hop >= min_hops and (dest is None or dest == nb) and not label_missing and labels_ok(nb)
            will_continue = max_hops is None or hop < max_hops
            if not will_emit and not will_continue:
                continue
            nwalk = walk + [e, nb] if emit_path else walk
            if will_emit:
                emit(nb, nwalk)
            if will_continue:
                stack.append((nb, nwalk, used + (e,), hop))
    return out


def trail_counts(edges, src, k):
    """Trails (edge-unique paths) of exactly k hops from `src`, counted per destination by depth-first enumeration —
    what CondVarLenTraverseOp's DFS (cond_var_len_traverse.rs:196-387) emits one row each for.  `edges` = list of
    (edge id, from, to) of the effective graph (multi-edges = several ids on one pair).  Small graphs only."""
    out_edges = {}
    for e, a, b in edges:
        out_edges.setdefault(a, []).append((e, b))
    counts = {}

    def dfs(v, depth, used):
        if depth == k:
            counts[v] = counts.get(v, 0) + 1
            return
        for e, w in out_edges.get(v, ()):
            if e in used:
                continue
            used.add(e)
            dfs(w, depth + 1, used)
            used.discard(e)

    dfs(src, 0, set())
    return counts


def expand_into_row(g: Graph, src, dst, types, bidirectional=False, emit_relationship=True, used_edges=()):
    """ExpandIntoOp::expand_row (expand_into.rs:121-258), SURVEY Appendix A.5: edge ids connecting the
    two bound endpoints, scanning the type tensors in order, ids ascending per type; without
    emit_relationship only the first surviving id per (src, dst) pair is kept."""
    pairs = [(src, dst)]
    if bidirectional and src != dst:
        pairs.append((dst, src))
    out = []
    tids = [g.type_ids[t] for t in types if t in g.type_ids] if types else list(range(len(g.tensors)))
    for (s, d) in pairs:
        got = []
        for t in tids:
            for e in g.tensors[t].get(s, d):
                if e in used_edges:
                    continue
                got.append((s, d, e))
        if not emit_relationship:
            got = got[:1]
        out.extend(got)
    return out


def algo_bfs(g: Graph, source, max_depth=-1, rel_type=None, want_edges=False):
    """algo.BFS (algo_procedures.rs:1021-1160), SURVEY Appendix A.6.  Returns None for "no row",
    else (nodes, edges): nodes ascending by id, excluding the source and deleted nodes."""
    if source is None or g.n == 0:
        return None
    if source in g.deleted_nodes:
        raise ValueError("Source node not found in graph")
    types = [rel_type] if rel_type is not None else []
    adj = g.build_adjacency_matrix(types)
    level, parent, _ = _bfs(adj, source, -1 if max_depth < 0 else max_depth, want_parent=True)
    nodes, edges = [], []
    for v in range(g.n):
        if level[v] < 0 or v == source or v in g.deleted_nodes:
            continue
        if want_edges:
            p = int(parent[v])
            if p in g.deleted_nodes:
                continue
            nodes.append(v)
            ids = g.get_src_dest_relationships(p, v, types)
            if ids:
                edges.append(ids[0])
        else:
            nodes.append(v)
    if not nodes:
        return None
    return nodes, edges


def algo_pagerank(g: Graph, label=None, rel_type=None):
    """algo.pageRank (algo_procedures.rs:687-783): (nodes ascending, scores) over the live nodes; a label that does
    not cover every live node selects the compact graph of its nodes (:711-733); deleted ids stay in the unfiltered
    matrix as isolated vertices (:718-720) and are dropped from the output (:768-770)."""
    from . import pagerank as _pr
    live = [v for v in range(g.n) if v not in g.deleted_nodes]
    if not live:
        return [], []
    types = [rel_type] if rel_type is not None else []
    active = None
    if label is not None:
        lids = g.resolve_label_ids([label])
        if lids is None:
            return [], []
        members = [v for v in live if g.node_has_label_id(v, lids[0])]
        if not members:
            return [], []
        if len(members) != len(live):
            active = np.zeros(g.n, dtype=bool)
            active[members] = True
    adj = g.build_adjacency_matrix(types)
    scores, _ = _pr.pagerank(adj, 0.85, 1e-4, 100, active=active)
    nodes = [v for v in live if active is None or active[v]]
    return nodes, [float(scores[v]) for v in nodes]


# ---- planner: fuse_anonymous_traverse (planner/optimizer/fuse_anonymous_traverse.rs) ------------------------
def _is_anon(alias):                                  # :38-40
    return alias.startswith("_anon")


def can_fuse(ops_by_id, parent, child):               # :83-188  (parent = outer hop (b)-->(c), child = (a)-->(b))
    p, c = ops_by_id[parent], ops_by_id[child]
    if p["kind"] != "CT" or c["kind"] != "CT":
        return False
    if not p.get("bind", True) or not c.get("bind", True):
        return False
    if p.get("optional") or c.get("optional"):        # :111-115
        return False
    if p.get("transposed") or c.get("transposed"):    # :118-122
        return False
    pr, cr = p["rel"], c["rel"]
    if not _is_anon(pr["alias"]) or not _is_anon(cr["alias"]):     # :125-127
        return False
    if p.get("emit") or c.get("emit"):                # :131-133
        return False
    if p.get("siblings") or c.get("siblings"):        # :135-137
        return False
    if pr.get("bidirectional") or cr.get("bidirectional"):         # :139-141
        return False
    if pr.get("var_len") or cr.get("var_len"):        # :142-144
        return False
    if pr.get("attrs") or cr.get("attrs"):            # :146-148
        return False
    if pr["from"]["alias"] != cr["to"]["alias"]:      # :150-154
        return False
    mid = pr["from"]
    if not _is_anon(mid["alias"]) or mid.get("labels") or mid.get("attrs"):   # :157-165
        return False
    cur = p["parent"]                                  # :66-79, :180-186: ancestors of the outer hop
    while cur >= 0:
        a = ops_by_id[cur]
        if a["kind"] == "X" and mid["alias"] in a.get("refs", []):
            return False
        cur = a["parent"]
    return True


def fuse_anonymous_traverse(ops):
    """ops: list of dicts (id, parent, kind "CT" | "X", ...), children ordered as listed.  Returns the list after the
    pass (:190-284): repeatedly merge the first fusable (parent, only-child) pair found in BFS order."""
    import copy
    ops = copy.deepcopy(ops)
    while True:
        by_id = {o["id"]: o for o in ops}
        kids = {o["id"]: [] for o in ops}
        root = None
        for o in ops:
            if o["parent"] < 0:
                root = o["id"]
            else:
                kids[o["parent"]].append(o["id"])
        target, queue = None, ([root] if root is not None else [])
        while queue and target is None:
            idx = queue.pop(0)
            queue.extend(kids[idx])
            if by_id[idx]["kind"] == "CT" and len(kids[idx]) == 1 and can_fuse(by_id, idx, kids[idx][0]):
                target = idx
        if target is None:
            return ops
        parent, child = by_id[target], by_id[kids[target][0]]
        parent["chain"] = list(child.get("chain", [])) + [parent["rel"]] + list(parent.get("chain", []))   # :236-241
        parent["rel"] = child["rel"]
        parent["transposed"], parent["optional"], parent["bind"] = False, False, True                   # :255-263
        for g in kids[child["id"]]:
            by_id[g]["parent"] = target
        # the grandchildren take the pruned child's place in the listing order
        pos = ops.index(child)
        ops.remove(child)
        grand = [o for o in ops if o["parent"] == target and o["id"] in kids[child["id"]]]
        for g in grand:
            ops.remove(g)
        for k, g in enumerate(grand):
            ops.insert(min(pos + k, len(ops)), g)
