"""CPU oracle for the FalkorDB traversal hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``falkordb_amd/`` does.  It restates, on the CPU, the semantics the
reference obtains from SuiteSparse:GraphBLAS v10.5.0 / LAGraph v1.3.x (un-vendored, absent from
/root/reference, not buildable here: no rustc, no libgraphblas), citing the reference call site
each function follows.  ``oracle.c`` holds the array kernels (gcc), this file the thin numpy
wrappers plus the pure-Python state machines (VersionedMatrix, Tensor read side, expand_batch,
ExpandInto, algo.BFS) that sit above them in the reference.

Parity pinning: see ``tests/test_oracle_golden.py`` — every known-answer the reference's own
tests hold for this path is replayed against this oracle.  Where the reference holds no
fixture (large masked products, LAGraph's parent choice) parity is *unpinned* and defined
structurally.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

U64 = np.uint64
_p64 = ctypes.POINTER(ctypes.c_uint64)


_OMP_PATH = os.path.join(_HERE, "_build", "liboracle_omp.so")
_omp = None


def build(force: bool = False) -> str:
    # staleness by CONTENT (a snapshot copied to another box carries libraries whose mtimes say nothing about the sources)
    import hashlib
    h = hashlib.sha256()
    for src in ("oracle.c", "oracle_omp.c", "Makefile"):
        with open(os.path.join(_HERE, src), "rb") as f:
            h.update(f.read())
    digest = h.hexdigest()
    stamp = os.path.join(_HERE, "_build", "sources.sha")
    stale = force or not (os.path.exists(_LIB_PATH) and os.path.exists(_OMP_PATH) and os.path.exists(stamp))
    if not stale:
        with open(stamp) as f:
            stale = f.read().strip() != digest
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(digest)
    return _LIB_PATH


def omp_lib():
    global _omp
    if _omp is None:
        build()
        L = ctypes.CDLL(_OMP_PATH)
        L.orc_bfs_omp.restype = ctypes.c_uint64
        L.orc_omp_threads.restype = ctypes.c_int
        L.orc_mxm_omp.restype = ctypes.c_uint64
        L.orc_merge_omp.restype = ctypes.c_uint64
        L.orc_checksum_omp.restype = ctypes.c_uint64
        L.orc_free.restype = None
        L.orc_free.argtypes = [ctypes.c_void_p]
        _omp = L
    return _omp


def omp_threads() -> int:
    return int(omp_lib().orc_omp_threads())


def bfs_omp(a: "CSR", at: "CSR | None", src: int, max_level: int = -1, threads: int = 0, alpha: float = 15.0):
    """Multi-threaded CPU stand-in for LAGraph's push/pull BFS (oracle_omp.c): bench.py's cpu_baseline.
    Returns (level int32[n], edges_traversed)."""
    level = np.zeros(a.nrows, dtype=np.int32)
    e = omp_lib().orc_bfs_omp(ctypes.c_uint64(a.nrows), _ptr(a.rowptr), _ptr(a.colidx),
                              _ptr(at.rowptr) if at is not None else None,
                              _ptr(at.colidx) if at is not None else None, ctypes.c_uint64(src),
                              ctypes.c_int64(max_level), level.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                              ctypes.c_int(threads), ctypes.c_double(alpha))
    return level, int(e)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_build_csr.restype = ctypes.c_uint64
        L.orc_mxm.restype = ctypes.c_uint64
        L.orc_merge.restype = ctypes.c_uint64
        L.orc_bfs.restype = ctypes.c_uint64
        L.orc_checksum.restype = ctypes.c_uint64
        L.orc_transpose.restype = None
        L.orc_vxm.restype = None
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=U64)


def _ptr(x):
    return x.ctypes.data_as(_p64) if x is not None else None


class CSR:
    """Sorted-unique CSR pattern (the wait()ed state of a reference Matrix<bool>)."""

    def __init__(self, nrows, ncols, rowptr, colidx):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.rowptr, self.colidx = _a(rowptr), _a(colidx)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def row(self, r):
        return self.colidx[int(self.rowptr[r]):int(self.rowptr[r + 1])]

    def pairs(self):
        rows = np.repeat(np.arange(self.nrows, dtype=U64), np.diff(self.rowptr).astype(np.int64))
        return rows, self.colidx.copy()

    def to_set(self):
        r, c = self.pairs()
        return set(zip(r.tolist(), c.tolist()))

    def __eq__(self, o):
        return (self.nrows == o.nrows and self.ncols == o.ncols and np.array_equal(self.rowptr, o.rowptr)
                and np.array_equal(self.colidx, o.colidx))


def build_csr(nrows, ncols, rows, cols) -> CSR:
    """Matrix::<bool>::build (matrix.rs:1281-1303): duplicate coordinates collapse."""
    rows, cols = _a(rows), _a(cols)
    n = len(rows)
    rp = np.zeros(nrows + 1, dtype=U64)
    ci = np.zeros(max(n, 1), dtype=U64)
    nnz = lib().orc_build_csr(ctypes.c_uint64(nrows), _ptr(rows), _ptr(cols), ctypes.c_uint64(n), _ptr(rp), _ptr(ci))
    return CSR(nrows, ncols, rp, ci[:nnz])


def empty(nrows, ncols) -> CSR:
    return CSR(nrows, ncols, np.zeros(nrows + 1, dtype=U64), np.zeros(0, dtype=U64))


def transpose(a: CSR) -> CSR:
    """Matrix::transpose (matrix.rs:633-662)."""
    trp = np.zeros(a.ncols + 1, dtype=U64)
    tci = np.zeros(max(a.nnz, 1), dtype=U64)
    lib().orc_transpose(ctypes.c_uint64(a.nrows), ctypes.c_uint64(a.ncols), _ptr(a.rowptr), _ptr(a.colidx),
                        _ptr(trp), _ptr(tci))
    return CSR(a.ncols, a.nrows, trp, tci[:a.nnz])


def mxm(f: CSR, b: CSR):
    """Matrix::lmxm (matrix.rs:930-947): C = F x B, ANY_PAIR, no mask.  Returns (C, flops)."""
    assert f.ncols == b.nrows
    crp = np.zeros(f.nrows + 1, dtype=U64)
    fl = ctypes.c_uint64(0)
    nnz = lib().orc_mxm(ctypes.c_uint64(f.nrows), _ptr(f.rowptr), _ptr(f.colidx), _ptr(b.rowptr), _ptr(b.colidx),
                        ctypes.c_uint64(b.ncols), _ptr(crp), None, ctypes.byref(fl))
    cci = np.zeros(max(nnz, 1), dtype=U64)
    lib().orc_mxm(ctypes.c_uint64(f.nrows), _ptr(f.rowptr), _ptr(f.colidx), _ptr(b.rowptr), _ptr(b.colidx),
                  ctypes.c_uint64(b.ncols), _ptr(crp), _ptr(cci), ctypes.byref(fl))
    return CSR(f.nrows, b.ncols, crp, cci[:nnz]), int(fl.value)


def merge(a: CSR, add: CSR | None, mask: CSR | None, mask_covers_add: bool = False) -> CSR:
    """(a \\ mask) U add row-wise: delta_lmxm's closing steps (matrix.rs:1382-1400) and
    VersionedMatrix::extract / flush (versioned_matrix.rs:609-620, 892-938)."""
    orp = np.zeros(a.nrows + 1, dtype=U64)
    cap = a.nnz + (add.nnz if add is not None else 0)
    oci = np.zeros(max(cap, 1), dtype=U64)
    nnz = lib().orc_merge(ctypes.c_uint64(a.nrows), _ptr(a.rowptr), _ptr(a.colidx),
                          _ptr(add.rowptr) if add is not None else None,
                          _ptr(add.colidx) if add is not None else None,
                          _ptr(mask.rowptr) if mask is not None else None,
                          _ptr(mask.colidx) if mask is not None else None,
                          ctypes.c_int(1 if mask_covers_add else 0), _ptr(orp), _ptr(oci))
    return CSR(a.nrows, a.ncols, orp, oci[:nnz])


def delta_lmxm(f: CSR, m: CSR, dp: CSR | None, dm: CSR | None):
    """Matrix::delta_lmxm (matrix.rs:1317-1402), SURVEY Appendix A.1.  Returns (C, flops) with
    flops counted over the m and dp products (the traversed edges)."""
    dp_n = dp.nnz if dp is not None else 0
    dm_n = dm.nnz if dm is not None else 0
    if dp_n == 0 and dm_n == 0:
        return mxm(f, m)                      # matrix.rs:1333-1337
    mask = acc = None
    flops = 0
    if dm_n > 0:
        mk, _ = mxm(f, dm)                    # matrix.rs:1343-1361
        if mk.nnz > 0:
            mask = mk
    if dp_n > 0:
        ac, fl = mxm(f, dp)                   # matrix.rs:1363-1380
        flops += fl
        if ac.nnz > 0:
            acc = ac
    c, fl = mxm(f, m)                         # matrix.rs:1382-1396 (GrB_DESC_RSC when mask)
    flops += fl
    if mask is not None or acc is not None:
        c = merge(c, acc, mask, False)        # replace+complemented mask, then eWiseAdd (:1398-1400)
    return c, flops


def _take_malloced(ptr, n):
    """numpy copy of a malloc'ed u64 array returned by oracle_omp.c, then free it."""
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].copy() if n else np.zeros(0, dtype=U64)
    omp_lib().orc_free(ctypes.cast(ptr, ctypes.c_void_p))
    return out


def mxm_omp(f: CSR, b: CSR, threads: int = 0):
    """Matrix::lmxm (matrix.rs:930-947), row-parallel (oracle_omp.c orc_mxm_omp): the same result as mxm()
    (tests/test_oracle_golden.py holds them equal), usable at BASELINE sizes.  Returns (C, flops)."""
    assert f.ncols == b.nrows
    crp = np.zeros(f.nrows + 1, dtype=U64)
    fl = ctypes.c_uint64(0)
    cci = _p64()
    nnz = omp_lib().orc_mxm_omp(ctypes.c_uint64(f.nrows), _ptr(f.rowptr), _ptr(f.colidx), _ptr(b.rowptr),
                                _ptr(b.colidx), ctypes.c_uint64(b.ncols), _ptr(crp), ctypes.byref(cci),
                                ctypes.byref(fl), ctypes.c_int(threads))
    return CSR(f.nrows, b.ncols, crp, _take_malloced(cci, nnz)), int(fl.value)


def merge_omp(a: CSR, add: CSR | None, mask: CSR | None, threads: int = 0) -> CSR:
    """(a \\ mask) U add, add entries not masked (matrix.rs:1382-1400), row-parallel."""
    orp = np.zeros(a.nrows + 1, dtype=U64)
    oci = _p64()
    nnz = omp_lib().orc_merge_omp(ctypes.c_uint64(a.nrows), _ptr(a.rowptr), _ptr(a.colidx),
                                  _ptr(add.rowptr) if add is not None else None,
                                  _ptr(add.colidx) if add is not None else None,
                                  _ptr(mask.rowptr) if mask is not None else None,
                                  _ptr(mask.colidx) if mask is not None else None, _ptr(orp), ctypes.byref(oci),
                                  ctypes.c_int(threads))
    return CSR(a.nrows, a.ncols, orp, _take_malloced(oci, nnz))


def delta_lmxm_omp(f: CSR, m: CSR, dp: CSR | None, dm: CSR | None, threads: int = 0):
    """Matrix::delta_lmxm (matrix.rs:1317-1402) exactly as delta_lmxm() above, on the row-parallel kernels."""
    dp_n = dp.nnz if dp is not None else 0
    dm_n = dm.nnz if dm is not None else 0
    if dp_n == 0 and dm_n == 0:
        return mxm_omp(f, m, threads)
    mask = acc = None
    flops = 0
    if dm_n > 0:
        mk, _ = mxm_omp(f, dm, threads)
        if mk.nnz > 0:
            mask = mk
    if dp_n > 0:
        ac, fl = mxm_omp(f, dp, threads)
        flops += fl
        if ac.nnz > 0:
            acc = ac
    c, fl = mxm_omp(f, m, threads)
    flops += fl
    if mask is not None or acc is not None:
        c = merge_omp(c, acc, mask, threads)
    return c, flops


def expand_omp(src_ids, layers, threads: int = 0, n: int | None = None):
    """The device core of expand_batch (cond_traverse.rs:600-605): F[i, src_i] = 1, then one delta_lmxm per hop.
    layers = [(m, dp, dm), ...].  Returns (C, flops, per-hop nnz)."""
    src_ids = np.asarray(src_ids, dtype=U64)
    n = n if n is not None else layers[0][0].nrows
    f = build_csr(len(src_ids), n, np.arange(len(src_ids), dtype=U64), src_ids)
    flops, hop_nnz = 0, []
    for (m, dp, dm) in layers:
        f, fl = delta_lmxm_omp(f, m, dp, dm, threads)
        flops += fl
        hop_nnz.append(f.nnz)
    return f, flops, hop_nnz


def expand_summary_omp(src_ids, layers, chunk: int = 64, threads: int = 0, n: int | None = None, seconds: float | None = None,
                       per_chunk: list | None = None):
    """(nnz, checksum, flops, per-hop nnz) of expand_omp() for result sets too large to hold at once (RMAT-24, 1024
    sources: 1.5 G entries): the sources go through the same chain `chunk` rows at a time — F keeps all len(src_ids)
    rows, only the chunk's rows are non-empty, so every (row, dest) pair is hashed with its row index in the whole
    batch and the partial checksums simply add (the checksum is a sum over the result set, oracle.c orc_checksum).
    With `seconds`, stops after the first chunk that crosses the budget and also returns how many sources were done
    (bench.py's bounded CPU baseline); otherwise returns the 4-tuple for the whole batch.  `per_chunk`, if a list, receives
    one (flops, seconds) pair per chunk (the quartiles of bench.py's cpu_baseline)."""
    src_ids = np.asarray(src_ids, dtype=U64)
    n = n if n is not None else layers[0][0].nrows
    import time as _time
    nnz = cs = flops = 0
    hop_nnz = [0] * len(layers)
    done, t0 = 0, _time.perf_counter()
    while done < len(src_ids):
        hi = min(done + chunk, len(src_ids))
        tc, fc = _time.perf_counter(), flops
        f = build_csr(len(src_ids), n, np.arange(done, hi, dtype=U64), src_ids[done:hi])
        for k, (m, dp, dm) in enumerate(layers):
            f, fl = delta_lmxm_omp(f, m, dp, dm, threads)
            flops += fl
            hop_nnz[k] += f.nnz
        nnz += f.nnz
        cs = (cs + checksum_omp(f, threads)) & 0xFFFFFFFFFFFFFFFF
        done = hi
        if per_chunk is not None:
            per_chunk.append((flops - fc, _time.perf_counter() - tc))
        if seconds is not None and _time.perf_counter() - t0 > seconds:
            break
    if seconds is not None:
        return nnz, cs, flops, hop_nnz, done, _time.perf_counter() - t0
    return nnz, cs, flops, hop_nnz


def checksum_omp(c: CSR, threads: int = 0) -> int:
    return int(omp_lib().orc_checksum_omp(ctypes.c_uint64(c.nrows), _ptr(c.rowptr), _ptr(c.colidx),
                                          ctypes.c_int(threads)))


def sample(a: CSR, seed: int, denom: int) -> CSR:
    """The entries fgpu_mat_sample keeps: mix64(seed ^ mix64(r << 32 | c)) % denom == 0."""
    r, c = a.pairs()
    h = mix64(np.uint64(seed) ^ mix64((r << np.uint64(32)) | c))
    keep = (h % np.uint64(denom)) == 0
    return build_csr(a.nrows, a.ncols, r[keep], c[keep])


def bfs(a: CSR, src: int, max_level: int = -1, want_parent: bool = True):
    """LAGr_BreadthFirstSearch_Extended contract as used by algo.BFS (algo_procedures.rs:1079-1160).
    Returns (level int32[n], parent int64[n] | None, edges_traversed)."""
    level = np.zeros(a.nrows, dtype=np.int32)
    parent = np.zeros(a.nrows, dtype=np.int64) if want_parent else None
    e = lib().orc_bfs(ctypes.c_uint64(a.nrows), _ptr(a.rowptr), _ptr(a.colidx), ctypes.c_uint64(src),
                      ctypes.c_int64(max_level), level.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                      parent.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)) if want_parent else None)
    return level, parent, int(e)


def vxm(a: CSR, f_bits: np.ndarray, mask_bits: np.ndarray | None) -> np.ndarray:
    """w<!mask,replace> = f x A, bitmap vectors (GrB_vxm, graphblas/mod.rs:11173)."""
    nw = (a.nrows + 63) // 64
    w = np.zeros(nw, dtype=U64)
    f_bits = _a(f_bits)
    m = _a(mask_bits) if mask_bits is not None else None
    lib().orc_vxm(ctypes.c_uint64(a.nrows), _ptr(a.rowptr), _ptr(a.colidx), _ptr(f_bits), _ptr(m), _ptr(w))
    return w


def checksum(c: CSR) -> int:
    return int(lib().orc_checksum(ctypes.c_uint64(c.nrows), _ptr(c.rowptr), _ptr(c.colidx)))


def bits_from_ids(n, ids) -> np.ndarray:
    w = np.zeros((n + 63) // 64, dtype=U64)
    ids = np.asarray(ids, dtype=np.uint64)
    np.bitwise_or.at(w, (ids >> np.uint64(6)).astype(np.int64), np.uint64(1) << (ids & np.uint64(63)))
    return w


def ids_from_bits(bits, n) -> np.ndarray:
    b = np.unpackbits(np.ascontiguousarray(bits, dtype=U64).view(np.uint8), bitorder="little")[:n]
    return np.nonzero(b)[0].astype(np.uint64)


# ---------------------------------------------------------------------------------------
# synthetic R-MAT — bit-identical to fgpu_mat_rmat's device generator
# ---------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    z = (np.asarray(z, dtype=U64) + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def rmat_scramble(x, scale):
    mask = np.uint32((1 << scale) - 1) if scale < 32 else np.uint32(0xFFFFFFFF)
    sh = np.uint32(scale // 2 + 1)
    x = np.asarray(x, dtype=np.uint32)
    x = (x * np.uint32(0x9E3779B1) + np.uint32(0x7F4A7C15)) & mask
    x = x ^ (x >> sh)
    x = (x * np.uint32(0x85EBCA6B)) & mask
    x = x ^ (x >> sh)
    x = (x * np.uint32(0xC2B2AE35) + np.uint32(0x165667B1)) & mask
    return x


def rmat_edges(scale, edge_factor=16, seed=None, a16=37356, b16=12452, c16=12452):
    """Raw (u, v) edge list incl. duplicates, self-loops removed — same stream as rmat_kernel."""
    if seed is None:
        seed = 0x5EED1234 + scale
    n_edges = edge_factor << scale
    i = np.arange(n_edges, dtype=U64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        a32 = min(a16 << 16, 0xFFFFFFFF)
        ab32 = min((a16 + b16) << 16, 0xFFFFFFFF)
        abc32 = min((a16 + b16 + c16) << 16, 0xFFFFFFFF)
        u = np.zeros(n_edges, dtype=np.uint32)
        v = np.zeros(n_edges, dtype=np.uint32)
        for l in range(scale):
            r = (mix64(base + np.uint64(((l + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)) >> np.uint64(32)).astype(np.uint64)
            ub = (r >= ab32).astype(np.uint32)
            vb = (((r >= a32) & (r < ab32)) | (r >= abc32)).astype(np.uint32)
            u = (u << np.uint32(1)) | ub
            v = (v << np.uint32(1)) | vb
        u = rmat_scramble(u, scale)
        v = rmat_scramble(v, scale)
    keep = u != v
    return u[keep].astype(U64), v[keep].astype(U64)


def rmat_csr(scale, edge_factor=16, seed=None) -> CSR:
    u, v = rmat_edges(scale, edge_factor, seed)
    n = 1 << scale
    return build_csr(n, n, u, v)
