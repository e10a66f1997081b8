"""ctypes handles over libfalkor_host.so (include/falkor_host.h) — the C++ host layer that mirrors
FalkorDB's Matrix / VersionedMatrix / Tensor / Graph slice and the CondTraverse / ExpandInto / algo.BFS
operators above the device C ABI.  Test and embedding plumbing only; nothing is computed here."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfalkor_host.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "falkor_host.h")
_lib = None

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


class HostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"falkor_host error {code}: {msg}")
        self.code = code


def declared_symbols():
    """Function names declared in include/falkor_host.h."""
    with open(HEADER) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", src)))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            from . import build
            build.build_host()
        L = C.CDLL(LIB_PATH)
        L.fh_last_error.restype = C.c_char_p
        L.fh_free.argtypes = [C.c_void_p]
        L.fh_free.restype = None
        for name in declared_symbols():
            fn = getattr(L, name)  # raises if the library lacks a declared symbol
            if name not in ("fh_last_error", "fh_free", "fh_finalize", "fh_mat_free", "fh_vm_free", "fh_graph_free",
                            "fh_last_op_ns", "fh_mat_cursor_free", "fh_tn_free"):
                fn.restype = C.c_int
        for name in ("fh_finalize", "fh_mat_free", "fh_vm_free", "fh_graph_free", "fh_mat_cursor_free", "fh_tn_free"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p]
        L.fh_last_op_ns.restype = C.c_uint64
        _lib = L
    return _lib


def _ck(code, allow=(0,)):
    if code not in allow:
        raise HostError(code, (load().fh_last_error() or b"").decode())
    return code


def _u64(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(u64p)


def _take(ptr, n, dtype=np.uint64):
    L = load()
    if not ptr:
        return np.zeros(0, dtype=dtype)
    out = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dtype, copy=True)
    L.fh_free(C.cast(ptr, C.c_void_p))
    return out


def should_fold(d, tx, base):
    return bool(load().fh_should_fold(C.c_uint64(d), C.c_uint64(tx), C.c_uint64(base)))


def should_fold_read(d, tx, base):
    return bool(load().fh_should_fold_read(C.c_uint64(d), C.c_uint64(tx), C.c_uint64(base)))


def delta_dominates_base(d, base):
    return bool(load().fh_delta_dominates_base(C.c_uint64(d), C.c_uint64(base)))


def compound_key(src, dst):
    k = C.c_uint64()
    _ck(load().fh_compound_key(C.c_uint64(src), C.c_uint64(dst), C.byref(k)))
    return k.value


class Context:
    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p()
        _ck(self.L.fh_init(C.byref(self.h), device))

    def close(self):
        if self.h:
            self.L.fh_finalize(self.h)
            self.h = C.c_void_p()


class MatrixCursor:
    """fh_mat_cursor_*: Iter::new / seek / next (matrix.rs:1471-1605); keeps its matrix alive."""

    def __init__(self, m, min_row, max_row, batch=4096):
        self.m, self.L, self.batch = m, m.L, batch
        self.h = C.c_void_p()
        _ck(self.L.fh_mat_cursor_new(m.h, C.c_uint64(min_row), C.c_uint64(max_row), C.byref(self.h)))

    def seek(self, min_row, max_row):
        _ck(self.L.fh_mat_cursor_seek(self.h, C.c_uint64(min_row), C.c_uint64(max_row)))
        return self

    def __iter__(self):
        r, c, v = (np.zeros(self.batch, dtype=np.uint64) for _ in range(3))
        n = C.c_uint64()
        while True:
            _ck(self.L.fh_mat_cursor_next(self.h, C.c_uint64(self.batch), _p(r), _p(c), _p(v), C.byref(n)))
            yield from zip(r[:n.value].tolist(), c[:n.value].tolist(), v[:n.value].tolist())
            if n.value < self.batch:
                return

    def __del__(self):
        try:
            if self.h:
                self.L.fh_mat_cursor_free(self.h)
        except Exception:
            pass


class Matrix:
    BOOL, UINT64 = 0, 1

    def __init__(self, ctx, typ=0, nrows=0, ncols=0, _h=None):
        self.ctx, self.L = ctx, ctx.L
        self.h = _h if _h is not None else C.c_void_p()
        if _h is None:
            _ck(self.L.fh_mat_new(ctx.h, C.byref(self.h), typ, C.c_uint64(nrows), C.c_uint64(ncols)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.fh_mat_free(self.h)
        except Exception:
            pass

    def _wrap(self, h):
        return Matrix(self.ctx, _h=h)

    def build(self, rows, cols, vals=None):
        r, c = _u64(rows), _u64(cols)
        v = _u64(vals) if vals is not None else None
        _ck(self.L.fh_mat_build(self.h, _p(r), _p(c), _p(v) if v is not None else None, C.c_uint64(len(r))))

    def set(self, i, j, v=1):
        _ck(self.L.fh_mat_set(self.h, C.c_uint64(i), C.c_uint64(j), C.c_uint64(v)))

    def remove(self, i, j):
        _ck(self.L.fh_mat_remove(self.h, C.c_uint64(i), C.c_uint64(j)))

    def get(self, i, j):
        v = C.c_uint64()
        code = _ck(self.L.fh_mat_get(self.h, C.c_uint64(i), C.c_uint64(j), C.byref(v)), allow=(0, 1))
        return None if code == 1 else v.value

    def nvals(self):
        v = C.c_uint64()
        _ck(self.L.fh_mat_nvals(self.h, C.byref(v)))
        return v.value

    def dims(self):
        a, b = C.c_uint64(), C.c_uint64()
        _ck(self.L.fh_mat_dims(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pending(self):
        v = C.c_int()
        _ck(self.L.fh_mat_pending(self.h, C.byref(v)))
        return bool(v.value)

    def wait(self):
        _ck(self.L.fh_mat_wait(self.h))

    def iter(self, min_row=0, max_row=2**64 - 1):
        r, c, v = u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_mat_iter(self.h, C.c_uint64(min_row), C.c_uint64(max_row), C.byref(r), C.byref(c), C.byref(v),
                               C.byref(n)))
        return list(zip(_take(r, n.value).tolist(), _take(c, n.value).tolist(), _take(v, n.value).tolist()))

    def cursor(self, min_row=0, max_row=2**64 - 1):
        """matrix::Iter as a reusable streaming cursor: .seek(min, max), iteration yields (row, col, val)."""
        return MatrixCursor(self, min_row, max_row)

    def _unary(self, fn, *args):
        h = C.c_void_p()
        _ck(fn(self.h, *args, C.byref(h)))
        return self._wrap(h)

    def dup(self):
        return self._unary(self.L.fh_mat_dup)

    def transpose(self):
        return self._unary(self.L.fh_mat_transpose)

    def grown(self, nrows, ncols):
        return self._unary(self.L.fh_mat_grown, C.c_uint64(nrows), C.c_uint64(ncols))

    def resize(self, nrows, ncols):
        _ck(self.L.fh_mat_resize(self.h, C.c_uint64(nrows), C.c_uint64(ncols)))

    def lmxm(self, b):
        _ck(self.L.fh_mat_lmxm(self.h, b.h))

    def rmxm(self, b):
        _ck(self.L.fh_mat_rmxm(self.h, b.h))

    def delta_lmxm(self, m, dp, dm):
        _ck(self.L.fh_mat_delta_lmxm(self.h, m.h, dp.h, dm.h))

    def intersection_nvals(self, b):
        v = C.c_uint64()
        _ck(self.L.fh_mat_intersection_nvals(self.h, b.h, C.byref(v)))
        return v.value


class VersionedMatrix:
    def __init__(self, ctx, nrows=0, ncols=0, _h=None):
        self.ctx, self.L = ctx, ctx.L
        self.h = _h if _h is not None else C.c_void_p()
        if _h is None:
            _ck(self.L.fh_vm_new(ctx.h, C.byref(self.h), C.c_uint64(nrows), C.c_uint64(ncols)))

    @classmethod
    def from_coo(cls, ctx, nrows, ncols, rows, cols):
        r, c = _u64(rows), _u64(cols)
        h = C.c_void_p()
        _ck(ctx.L.fh_vm_from_coo(ctx.h, C.byref(h), C.c_uint64(nrows), C.c_uint64(ncols), _p(r), _p(c),
                                 C.c_uint64(len(r))))
        return cls(ctx, _h=h)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.fh_vm_free(self.h)
        except Exception:
            pass

    def set(self, i, j, _v=True):
        _ck(self.L.fh_vm_set(self.h, C.c_uint64(i), C.c_uint64(j)))

    def remove(self, i, j):
        _ck(self.L.fh_vm_remove(self.h, C.c_uint64(i), C.c_uint64(j)))

    def get(self, i, j):
        return True if _ck(self.L.fh_vm_get(self.h, C.c_uint64(i), C.c_uint64(j)), allow=(0, 1)) == 0 else None

    def nvals(self):
        v = C.c_uint64()
        _ck(self.L.fh_vm_nvals(self.h, C.byref(v)))
        return v.value

    def iter(self, min_row=0, max_row=2**64 - 1):
        r, c = u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_vm_iter(self.h, C.c_uint64(min_row), C.c_uint64(max_row), C.byref(r), C.byref(c), C.byref(n)))
        return list(zip(_take(r, n.value).tolist(), _take(c, n.value).tolist()))

    def set_all(self, entries, new=False):
        e = list(entries)
        r, c = _u64([x[0] for x in e]), _u64([x[1] for x in e])
        _ck(self.L.fh_vm_set_all(self.h, _p(r), _p(c), C.c_uint64(len(e)), 1 if new else 0))

    def remove_mask(self, entries):
        e = list(entries)
        r, c = _u64([x[0] for x in e]), _u64([x[1] for x in e])
        _ck(self.L.fh_vm_remove_mask(self.h, _p(r), _p(c), C.c_uint64(len(e))))

    def dup(self):
        h = C.c_void_p()
        _ck(self.L.fh_vm_dup(self.h, C.byref(h)))
        return VersionedMatrix(self.ctx, _h=h)

    def transpose(self):
        h = C.c_void_p()
        _ck(self.L.fh_vm_transpose(self.h, C.byref(h)))
        return VersionedMatrix(self.ctx, _h=h)

    def wait(self):
        _ck(self.L.fh_vm_wait(self.h))

    def flush(self):
        _ck(self.L.fh_vm_flush(self.h))

    def fold_oversized(self):
        _ck(self.L.fh_vm_fold_oversized(self.h))

    def extract(self):
        h = C.c_void_p()
        _ck(self.L.fh_vm_extract(self.h, C.byref(h)))
        return Matrix(self.ctx, _h=h)

    def state(self):
        s = (C.c_uint64 * 4)()
        _ck(self.L.fh_vm_state(self.h, s))
        return {"m": s[0], "dp": s[1], "dm": s[2], "needs_flush": bool(s[3])}


def cond_spec(src_labels=(), hops=((), ()), optional=False, bind=False, emit=False, bidir=False, siblings=False,
              attrs=False, transposed=False):
    """hops: sequence of (types, dst_labels); hop 0 is the operator's own pattern, the rest the fused chain."""
    parts = ["src=" + ",".join(src_labels)]
    for types, labels in hops:
        parts.append("hop=" + ",".join(types) + "|" + ",".join(labels))
    for k, v in (("optional", optional), ("bind", bind), ("emit", emit), ("bidir", bidir), ("siblings", siblings),
                 ("attrs", attrs), ("transposed", transposed)):
        parts.append(f"{k}={1 if v else 0}")
    return ";".join(parts).encode()


class Graph:
    def __init__(self, ctx, node_cap):
        self.ctx, self.L = ctx, ctx.L
        self.h = C.c_void_p()
        self.n = node_cap
        _ck(self.L.fh_graph_new(ctx.h, C.byref(self.h), C.c_uint64(node_cap)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.L.fh_graph_free(self.h)
        except Exception:
            pass

    def add_label(self, name):
        v = C.c_uint64()
        _ck(self.L.fh_graph_add_label(self.h, name.encode(), C.byref(v)))
        return v.value

    def add_type(self, name):
        v = C.c_uint64()
        _ck(self.L.fh_graph_add_type(self.h, name.encode(), C.byref(v)))
        return v.value

    def label_node(self, node, label_id):
        _ck(self.L.fh_graph_label_node(self.h, C.c_uint64(node), C.c_uint64(label_id)))

    def delete_node(self, node):
        _ck(self.L.fh_graph_delete_node(self.h, C.c_uint64(node)))

    def create_edge(self, type_id, src, dst, edge_id):
        _ck(self.L.fh_graph_create_edge(self.h, C.c_uint64(type_id), C.c_uint64(src), C.c_uint64(dst),
                                        C.c_uint64(edge_id)))

    def create_edges(self, type_id, srcs, dsts, ids):
        s, d, i = _u64(srcs), _u64(dsts), _u64(ids)
        _ck(self.L.fh_graph_create_edges(self.h, C.c_uint64(type_id), _p(s), _p(d), _p(i), C.c_uint64(len(s))))

    def delete_edge(self, type_id, src, dst, edge_id):
        _ck(self.L.fh_graph_delete_edge(self.h, C.c_uint64(type_id), C.c_uint64(src), C.c_uint64(dst),
                                        C.c_uint64(edge_id)))

    def commit(self):
        _ck(self.L.fh_graph_commit(self.h))

    def node_has_label(self, node, label_id):
        return _ck(self.L.fh_graph_node_has_label(self.h, C.c_uint64(node), C.c_uint64(label_id)), allow=(0, 1)) == 0

    def tensor_get(self, type_id, src, dst):
        p, n = u64p(), C.c_uint64()
        _ck(self.L.fh_tensor_get(self.h, C.c_uint64(type_id), C.c_uint64(src), C.c_uint64(dst), C.byref(p),
                                 C.byref(n)))
        return _take(p, n.value).tolist()

    def tensor_edge_count(self, type_id):
        v = C.c_uint64()
        _ck(self.L.fh_tensor_edge_count(self.h, C.c_uint64(type_id), C.byref(v)))
        return v.value

    def tensor_iter_edges(self, type_id):
        s, d, i = u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_tensor_iter_edges(self.h, C.c_uint64(type_id), C.byref(s), C.byref(d), C.byref(i), C.byref(n)))
        return list(zip(_take(s, n.value).tolist(), _take(d, n.value).tolist(), _take(i, n.value).tolist()))

    def tensor_state(self, type_id):
        s = (C.c_uint64 * 5)()
        _ck(self.L.fh_tensor_state(self.h, C.c_uint64(type_id), s))
        return {"m": s[0], "dp": s[1], "dm": s[2], "multi_pairs": s[3], "mt": s[4]}

    def layer(self, type_id, which):
        """Raw (row, col, val) entries of one layer: type_id None = adjacency; which in 'm', 'dp', 'dm'."""
        r, c, v = u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_graph_layer_iter(self.h, C.c_int64(-1 if type_id is None else type_id),
                                       {"m": 0, "dp": 1, "dm": 2}[which], C.byref(r), C.byref(c), C.byref(v),
                                       C.byref(n)))
        return list(zip(_take(r, n.value).tolist(), _take(c, n.value).tolist(), _take(v, n.value).tolist()))

    # ---- operators -----------------------------------------------------------------------------
    def cond_traverse_batch(self, spec, src, to_bound=None, as_arrays=False):
        """src / to_bound: node id, None = bound to NULL / non-node (src) or unbound (to_bound).
        Returns None when the batched path bails to the per-row one, else (rows, null_rows, flops) with
        rows = [(active_row, dest[, edge])]."""
        k = len(src)
        s = np.asarray([(-2 if v is None else v) for v in src], dtype=np.int64)
        tb = None if to_bound is None else np.asarray([(-1 if v is None else v) for v in to_bound], dtype=np.int64)
        batched = C.c_int()
        orow, odst, onull = u64p(), u64p(), u64p()
        oedge = i64p()
        n, nn, fl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _ck(self.L.fh_cond_traverse_batch(self.h, spec, s.ctypes.data_as(i64p),
                                          tb.ctypes.data_as(i64p) if tb is not None else None, C.c_uint64(k),
                                          C.byref(batched), C.byref(orow), C.byref(odst), C.byref(oedge), C.byref(n),
                                          C.byref(onull), C.byref(nn), C.byref(fl)))
        if as_arrays:   # numpy views of the result columns (no per-row Python objects)
            res = (_take(orow, n.value), _take(odst, n.value), _take(oedge, n.value, np.int64))
            nulls = _take(onull, nn.value)
            return (res, nulls, fl.value) if batched.value else None
        rows, dst = _take(orow, n.value).tolist(), _take(odst, n.value).tolist()
        edge = _take(oedge, n.value, np.int64).tolist()
        nulls = _take(onull, nn.value).tolist()
        if not batched.value:
            return None
        out = [(r, d) if e < 0 else (r, d, e) for r, d, e in zip(rows, dst, edge)]
        return out, nulls, fl.value

    def cond_traverse_row(self, spec, from_id=None, to_id=None, transposed=False):
        f, t, e = u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_cond_traverse_row(self.h, spec, C.c_int64(-1 if from_id is None else from_id),
                                        C.c_int64(-1 if to_id is None else to_id), 1 if transposed else 0,
                                        C.byref(f), C.byref(t), C.byref(e), C.byref(n)))
        return list(zip(_take(f, n.value).tolist(), _take(t, n.value).tolist(), _take(e, n.value).tolist()))

    def cond_traverse_rows(self, spec, from_ids, to_ids, transposed=False, used_edges=(), dedup_src=None):
        """The per-row fallback over one input batch: from / to = node id, None (unbound) or "x" (bound to a non-node).
        Returns [(row, from, to, edge)] in emission order."""
        enc = lambda v: -1 if v is None else (-2 if isinstance(v, str) else int(v))
        k = len(from_ids)
        f = (C.c_int64 * k)(*[enc(v) for v in from_ids])
        t = (C.c_int64 * k)(*[enc(v) for v in to_ids])
        dd = (C.c_int64 * k)(*[enc(v) for v in dedup_src]) if dedup_src is not None else None
        used = _u64(list(used_edges))
        orow, of, ot, oe = u64p(), u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_cond_traverse_rows(self.h, spec, f, t, dd, C.c_uint64(k), 1 if transposed else 0, _p(used),
                                         C.c_uint64(len(used)), C.byref(orow), C.byref(of), C.byref(ot), C.byref(oe),
                                         C.byref(n)))
        return list(zip(_take(orow, n.value).tolist(), _take(of, n.value).tolist(), _take(ot, n.value).tolist(),
                        _take(oe, n.value).tolist()))

    def expand_into(self, types, srcs, dsts, bidirectional=False, emit_relationship=True, batched=True):
        s, d = _u64(srcs), _u64(dsts)
        orow, osrc, odst, oedge = u64p(), u64p(), u64p(), u64p()
        n = C.c_uint64()
        _ck(self.L.fh_expand_into(self.h, ",".join(types).encode(), 1 if bidirectional else 0,
                                  1 if emit_relationship else 0, 1 if batched else 0, _p(s), _p(d),
                                  C.c_uint64(len(s)), C.byref(orow), C.byref(osrc), C.byref(odst), C.byref(oedge),
                                  C.byref(n)))
        return list(zip(_take(orow, n.value).tolist(), _take(osrc, n.value).tolist(),
                        _take(odst, n.value).tolist(), _take(oedge, n.value).tolist()))

    def var_len_traverse(self, start, types=(), dest=None, min_hops=1, max_hops=None, reversed=False,
                         bidirectional=False, dst_labels=(), emit_path=False, prune=True):
        """CondVarLenTraverse over one input row (cond_var_len_traverse.rs:81-387): [(from, to, path | None)] in the
        reference's emission order, plus {frames, pruned, reach_products}."""
        of, ot, op_, off = u64p(), u64p(), u64p(), u64p()
        n = C.c_uint64()
        st = (C.c_uint64 * 3)()
        _ck(self.L.fh_var_len_traverse(self.h, ",".join(types).encode(), ",".join(dst_labels).encode(),
                                       1 if reversed else 0, 1 if bidirectional else 0, C.c_uint32(min_hops),
                                       C.c_uint32(0xFFFFFFFF if max_hops is None else max_hops), C.c_uint64(start),
                                       C.c_int64(-1 if dest is None else dest), 1 if emit_path else 0, 1 if prune else 0,
                                       C.byref(of), C.byref(ot), C.byref(op_), C.byref(off), C.byref(n), st))
        k = n.value
        offs = _take(off, k + 1).tolist()
        f, t = _take(of, k).tolist(), _take(ot, k).tolist()
        p = _take(op_, offs[-1]).tolist()
        rows = [(f[i], t[i], p[offs[i]:offs[i + 1]] if emit_path else None) for i in range(k)]
        return rows, {"frames": int(st[0]), "pruned": int(st[1]), "reach_products": int(st[2])}

    def build_adjacency(self, types=(), symmetric=False):
        """Graph::build_adjacency_matrix / build_symmetric_adjacency_matrix (graph.rs:3870-3907) -> Matrix"""
        h = C.c_void_p()
        _ck(self.L.fh_graph_build_adjacency(self.h, ",".join(types).encode(), 1 if symmetric else 0, C.byref(h)))
        return Matrix(self.ctx, _h=h)

    def algo_pagerank(self, label=None, rel_type=None):
        """CALL algo.pageRank(label, relationshipType) YIELD node, score -> (nodes, scores float64)."""
        nodes = u64p()
        scores = C.POINTER(C.c_double)()
        n = C.c_uint64()
        _ck(self.L.fh_algo_pagerank(self.h, label.encode() if label is not None else None,
                                    rel_type.encode() if rel_type is not None else None, C.byref(nodes),
                                    C.byref(scores), C.byref(n)))
        return _take(nodes, n.value), _take(scores, n.value, np.float64)

    def algo_bfs(self, source, max_depth=-1, rel_type=None, want_edges=False):
        has = C.c_int()
        nodes, edges = u64p(), u64p()
        nn, ne = C.c_uint64(), C.c_uint64()
        _ck(self.L.fh_algo_bfs(self.h, C.c_int64(-1 if source is None else source), C.c_int64(max_depth),
                               rel_type.encode() if rel_type is not None else None, 1 if want_edges else 0,
                               C.byref(has), C.byref(nodes), C.byref(nn), C.byref(edges), C.byref(ne)))
        nv, ev = _take(nodes, nn.value).tolist(), _take(edges, ne.value).tolist()
        return (nv, ev) if has.value else None


class Tensor:
    """A Tensor on its own (tensor.rs:184-989), as the reference's unit tests drive it (fh_tn_*)."""
    MULTI_EDGE = 2 ** 64 - 1

    def __init__(self, ctx, nrows=None, ncols=None, _h=None):
        self.ctx, self.L = ctx, ctx.L
        self.h = _h if _h is not None else C.c_void_p()
        if _h is None:
            _ck(self.L.fh_tn_new(ctx.h, C.byref(self.h), C.c_uint64(nrows), C.c_uint64(ncols)))

    def __del__(self):
        try:
            if self.h and self.ctx.h:      # a tensor that outlives its context has nothing left to free
                self.L.fh_tn_free(self.h)
            self.h = C.c_void_p()
        except Exception:
            pass

    def dup(self):
        h = C.c_void_p()
        _ck(self.L.fh_tn_dup(self.h, C.byref(h)))
        return Tensor(self.ctx, _h=h)

    def set_all_from_slices(self, srcs, dsts, ids):
        s, d, i = _u64(srcs), _u64(dsts), _u64(ids)
        _ck(self.L.fh_tn_set_all(self.h, _p(s), _p(d), _p(i), C.c_uint64(len(s))))

    def remove_all(self, rels):
        """rels = [(edge id, src, dst)]; returns the emptied (src, dst) pairs."""
        flat = _u64(np.asarray(list(rels), dtype=np.uint64).reshape(-1))
        es, ed, n = u64p(), u64p(), C.c_uint64()
        _ck(self.L.fh_tn_remove_all(self.h, _p(flat), C.c_uint64(len(flat) // 3), C.byref(es), C.byref(ed), C.byref(n)))
        return list(zip(_take(es, n.value).tolist(), _take(ed, n.value).tolist()))

    def flush(self): _ck(self.L.fh_tn_op(self.h, 0, C.c_uint64(0), C.c_uint64(0)))
    def fold_oversized(self): _ck(self.L.fh_tn_op(self.h, 1, C.c_uint64(0), C.c_uint64(0)))
    def wait(self): _ck(self.L.fh_tn_op(self.h, 2, C.c_uint64(0), C.c_uint64(0)))
    def wait_fwd(self): _ck(self.L.fh_tn_op(self.h, 3, C.c_uint64(0), C.c_uint64(0)))
    def resize(self, nrows, ncols): _ck(self.L.fh_tn_op(self.h, 4, C.c_uint64(nrows), C.c_uint64(ncols)))

    def get(self, src, dst):
        ids, n = u64p(), C.c_uint64()
        _ck(self.L.fh_tn_get(self.h, C.c_uint64(src), C.c_uint64(dst), C.byref(ids), C.byref(n)))
        return _take(ids, n.value).tolist()

    def _probe(self, which, src, dst):
        v = C.c_uint64()
        code = self.L.fh_tn_probe(self.h, which, C.c_uint64(src), C.c_uint64(dst), C.byref(v))
        _ck(code, allow=(0, 1))
        return v.value if code == 0 else None

    def eff_get(self, src, dst): return self._probe(0, src, dst)
    def m_get(self, src, dst): return self._probe(1, src, dst)
    def extract_contains(self, src, dst): return self._probe(2, src, dst) is not None

    def encode(self) -> bytes:
        b, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        _ck(self.L.fh_tn_encode(self.h, C.byref(b), C.byref(n)))
        out = bytes(C.cast(b, C.POINTER(C.c_uint8 * max(n.value, 1))).contents)[:n.value]
        self.L.fh_free(b)
        return out

    @staticmethod
    def decode(ctx, payload: bytes):
        buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
        h, used = C.c_void_p(), C.c_uint64()
        _ck(ctx.L.fh_tn_decode(ctx.h, buf, C.c_uint64(len(payload)), C.byref(h), C.byref(used)))
        return Tensor(ctx, _h=h), used.value

    def state(self):
        out = (C.c_uint64 * 8)()
        _ck(self.L.fh_tn_state(self.h, out))
        return dict(zip(("m", "dp", "dm", "multi_pairs", "mt", "me_nvals", "edge_count", "m_pending"), [int(x) for x in out]))


def algo_bfs_multi(graph, gang, source, max_depth=-1, rel_type=None, want_edges=False):
    """algo.BFS partitioned over the contexts of `gang` (host.Context objects, gang[0] = the graph's own)."""
    has = C.c_int()
    nodes, edges = u64p(), u64p()
    nn, ne = C.c_uint64(), C.c_uint64()
    arr = (C.c_void_p * len(gang))(*[c.h for c in gang])
    _ck(graph.L.fh_algo_bfs_multi(graph.h, arr, len(gang), C.c_int64(-1 if source is None else source),
                                  C.c_int64(max_depth), rel_type.encode() if rel_type is not None else None,
                                  1 if want_edges else 0, C.byref(has), C.byref(nodes), C.byref(nn), C.byref(edges),
                                  C.byref(ne)))
    nv, ev = _take(nodes, nn.value).tolist(), _take(edges, ne.value).tolist()
    return (nv, ev) if has.value else None


# ---- v19 matrix payload (Encode<19> / Decode<19> for Matrix<T>) ---------------------------------------------
def container_parse(payload: bytes):
    """CPU-only: {nrows, ncols, nvals, hyper, valued, consumed, p, h, i, x} of a container payload."""
    L = load()
    buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
    dims = (C.c_uint64 * 6)()
    p, h, i, x = u64p(), u64p(), u64p(), u64p()
    np_, nh = C.c_uint64(), C.c_uint64()
    _ck(L.fh_container_parse(buf, C.c_uint64(len(payload)), dims, C.byref(p), C.byref(np_), C.byref(h), C.byref(nh),
                             C.byref(i), C.byref(x)))
    nvals, valued = dims[2], bool(dims[4])
    return {"nrows": dims[0], "ncols": dims[1], "nvals": nvals, "hyper": bool(dims[3]), "valued": valued,
            "consumed": dims[5], "p": _take(p, np_.value), "h": _take(h, nh.value), "i": _take(i, nvals),
            "x": _take(x, nvals if valued else 0)}


def matrix_decode(ctx, payload: bytes):
    """Decode<19> for Matrix<T>: payload -> Matrix (arrays go to the device as they are)."""
    buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
    h = C.c_void_p()
    used = C.c_uint64()
    _ck(ctx.L.fh_mat_decode(ctx.h, buf, C.c_uint64(len(payload)), C.byref(h), C.byref(used)))
    return Matrix(ctx, _h=h), used.value


def matrix_encode(m) -> bytes:
    """Encode<19> for Matrix<T>."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_uint64()
    _ck(m.L.fh_mat_encode(m.h, C.byref(out), C.byref(n)))
    data = bytes(out[:n.value])
    m.L.fh_free(C.cast(out, C.c_void_p))
    return data


# ---- planner slice (fuse_anonymous_traverse) -------------------------------------------------------------
def _node_txt(n):
    return n["alias"] + "".join(":" + l for l in n.get("labels", [])) + ("*" if n.get("attrs") else "")


def _rel_txt(r):
    flags = ("b" if r.get("bidirectional") else "") + ("v" if r.get("var_len") else "") + ("a" if r.get("attrs") else "")
    return "|".join([r["alias"], _node_txt(r["from"]), _node_txt(r["to"]), ",".join(r.get("types", [])), flags])


def plan_to_text(ops):
    """ops: list of dicts {id, parent, kind: "CT" | "X", ...} (see tests/test_host_cpu.py); children keep list order."""
    lines = []
    for op in ops:
        if op["kind"] == "X":
            lines.append(f"X {op['id']} {op['parent']} {op.get('name', 'Op')} refs={','.join(op.get('refs', []))}")
        else:
            flags = ("e" if op.get("emit") else "") + ("t" if op.get("transposed") else "") + \
                    ("o" if op.get("optional") else "") + ("" if op.get("bind", True) else "n")
            lines.append(f"CT {op['id']} {op['parent']} rel={_rel_txt(op['rel'])} flags={flags} "
                         f"sib={','.join(op.get('siblings', []))} chain={';'.join(_rel_txt(r) for r in op.get('chain', []))}")
    return "\n".join(lines) + "\n"


def _node_from(txt):
    attrs = txt.endswith("*")
    parts = txt.rstrip("*").split(":")
    return {"alias": parts[0], "labels": [p for p in parts[1:] if p], "attrs": attrs}


def _rel_from(txt):
    f = (txt.split("|") + [""] * 5)[:5]
    return {"alias": f[0], "from": _node_from(f[1]), "to": _node_from(f[2]), "types": [t for t in f[3].split(",") if t],
            "bidirectional": "b" in f[4], "var_len": "v" in f[4], "attrs": "a" in f[4]}


def plan_from_text(text):
    ops = []
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        kv = dict(t.split("=", 1) for t in tok[3:] if "=" in t)
        if tok[0] == "X":
            ops.append({"id": int(tok[1]), "parent": int(tok[2]), "kind": "X", "name": tok[3],
                        "refs": [a for a in kv.get("refs", "").split(",") if a]})
        else:
            fl = kv.get("flags", "")
            ops.append({"id": int(tok[1]), "parent": int(tok[2]), "kind": "CT", "rel": _rel_from(kv["rel"]),
                        "emit": "e" in fl, "transposed": "t" in fl, "optional": "o" in fl, "bind": "n" not in fl,
                        "siblings": [a for a in kv.get("sib", "").split(",") if a],
                        "chain": [_rel_from(r) for r in kv.get("chain", "").split(";") if r]})
    return ops


def plan_fuse(ops, lower_id=-1):
    """fuse_anonymous_traverse on a plan (list of op dicts); returns (ops after the pass, runtime spec of CondTraverse
    node `lower_id` as bytes for Graph.cond_traverse_batch, or None)."""
    L = load()
    out, spec = C.c_char_p(), C.c_char_p()
    L.fh_plan_fuse.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
    _ck(L.fh_plan_fuse(plan_to_text(ops).encode(), lower_id, C.byref(out), C.byref(spec)))
    text, sp = out.value.decode(), spec.value
    return plan_from_text(text), (sp if lower_id >= 0 else None)
