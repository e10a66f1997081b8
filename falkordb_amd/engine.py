"""Thin Python handles over the C ABI (include/fgpu.h) — test / bench plumbing only.

Everything of substance happens inside libfgpu.so; these classes own handles, convert numpy
arrays to the plain pointers the ABI takes, and raise FgpuError on any non-zero fgpu_info.
"""
from __future__ import annotations

import ctypes as C
import os

import weakref

import numpy as np

from . import _ffi
from ._ffi import FgpuError, check, u64p, u8p, i32p, i64p

U64 = np.uint64


def _u64(x):
    return np.ascontiguousarray(x, dtype=U64)


def _p(a, t=u64p):
    return a.ctypes.data_as(t) if a is not None else None


class Context:
    """fgpu_ctx: one per process+device (matrix::init analogue, matrix.rs:116-185)."""

    def __init__(self, device: int = 0):
        self.lib = _ffi.load()
        self._h = C.c_void_p()
        self._live_views = 0          # result arrays handed out as views of library memory (_take) and not yet dropped
        self._close_pending = False
        check(self.lib.fgpu_init(C.byref(self._h), device, None, None))
        # A/B plumbing for the tools and the test-suite: FGPU_OPTS="name=value,name=value" sets library options on every context this
        # process opens (e.g. the whole parity suite under an experiment switch) — the library itself reads no environment for them
        for kv in filter(None, os.environ.get("FGPU_OPTS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))

    @property
    def handle(self):
        return self._h

    def close(self):
        """fgpu_finalize.  Result arrays are zero-copy views of pinned blocks the context owns: while any of them is alive
        the finalize is put off until the last one is dropped (an array that outlived its context would dangle)."""
        if not self._h:
            return
        if self._live_views > 0:
            self._close_pending = True
            return
        self.lib.fgpu_finalize(self._h)
        self._h = C.c_void_p()
        self._close_pending = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.fgpu_sync(self._h))

    def set_stream(self, stream_ptr: int | None):
        check(self.lib.fgpu_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def set_option(self, name: str, value: int):
        check(self.lib.fgpu_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        check(self.lib.fgpu_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    # ---- multi-GPU communicator (RCCL inside libfgpu.so; dist.hip) ----
    def comm_unique_id(self) -> bytes:
        """ncclGetUniqueId: rank 0 calls it, the launcher's own channel carries the 128 bytes to the other ranks."""
        buf = (C.c_uint8 * 128)()
        check(self.lib.fgpu_comm_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, nranks: int, rank: int, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(self.lib.fgpu_comm_init_rank(self._h, nranks, rank, buf))

    def comm_finalize(self):
        check(self.lib.fgpu_comm_finalize(self._h))

    def comm_info(self):
        r, n = C.c_int32(), C.c_int32()
        check(self.lib.fgpu_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def prof_enable(self, enable=True):
        """Context-wide kernel profiler (HIP events around the modelled kernels of the non-BFS paths)."""
        check(self.lib.fgpu_prof_enable(self._h, 1 if enable else 0))

    def prof_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_uint64 * cap)()
        ab = (C.c_uint64 * cap)()
        n = C.c_int()
        check(self.lib.fgpu_prof_read(self._h, names, ms, launches, ab, cap, C.byref(n)))
        return [{"kernel": names[i].decode(), "ms": ms[i], "launches": int(launches[i]), "alg_bytes": int(ab[i])}
                for i in range(n.value)]

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, wave = C.c_int32(), C.c_int32()
        lds, hbm = C.c_int64(), C.c_int64()
        check(self.lib.fgpu_device_info(self._h, name, C.byref(cus), C.byref(wave), C.byref(lds), C.byref(hbm)))
        return {"name": name.value.decode(), "cus": cus.value, "wave": wave.value, "lds_bytes": lds.value,
                "hbm_bytes": hbm.value}

    def device_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(self.lib.fgpu_device_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- helpers to pull caller-owned host buffers back into numpy ----
    def _take(self, ptr, n, dtype=U64):
        """A numpy VIEW of a result array the library handed out (no copy: the arrays of a k-hop batch are hundreds of
        MB); fgpu_free runs when the array — and every view derived from it — is gone."""
        if not ptr or n == 0:
            if ptr:
                self.lib.fgpu_free(self._h, C.cast(ptr, C.c_void_p))
            return np.zeros(0, dtype=dtype)
        addr = C.cast(ptr, C.c_void_p).value
        buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(addr)
        arr = np.frombuffer(buf, dtype=dtype, count=int(n))
        ctx = self                               # (a strong reference: the context outlives its views)
        ctx._live_views += 1

        def release():
            ctx._live_views -= 1
            if ctx._h:
                ctx.lib.fgpu_free(ctx._h, C.c_void_p(addr))
                if ctx._close_pending and ctx._live_views == 0:
                    ctx.close()
        weakref.finalize(buf, release)
        return arr

    def host_array(self, n, dtype):
        """fgpu_host_alloc: a pinned host array from the context's pool, for outputs the caller provides (level[] /
        parent[] of fgpu_bfs): the library fills it by DMA instead of staging + a host copy."""
        p = C.c_void_p()
        check(self.lib.fgpu_host_alloc(self._h, int(n) * np.dtype(dtype).itemsize, C.byref(p)))
        return self._take(p, n, dtype)

    # ---- matrix factories ----
    def mat_new(self, nrows, ncols) -> "Mat":
        h = C.c_void_p()
        check(self.lib.fgpu_mat_new(self._h, C.byref(h), nrows, ncols))
        return Mat(self, h)

    def mat_from_coo(self, nrows, ncols, rows, cols, vals=None) -> "Mat":
        rows, cols = _u64(rows), _u64(cols)
        v = _u64(vals) if vals is not None else None
        h = C.c_void_p()
        check(self.lib.fgpu_mat_from_coo(self._h, C.byref(h), nrows, ncols, _p(rows), _p(cols), _p(v), len(rows)))
        return Mat(self, h)

    def mat_from_csr(self, nrows, ncols, rowptr, colidx, vals=None, hyper_rows=None) -> "Mat":
        rowptr, colidx = _u64(rowptr), _u64(colidx)
        v = _u64(vals) if vals is not None else None
        hr = _u64(hyper_rows) if hyper_rows is not None else None
        h = C.c_void_p()
        check(self.lib.fgpu_mat_from_csr(self._h, C.byref(h), nrows, ncols, len(colidx),
                                         rowptr.ctypes.data_as(C.c_void_p), 64, colidx.ctypes.data_as(C.c_void_p), 64,
                                         _p(v), _p(hr), len(hr) if hr is not None else 0))
        return Mat(self, h)

    def mat_rmat(self, scale, edge_factor=16, seed=None, a16=0, b16=0, c16=0) -> "Mat":
        if seed is None:
            seed = 0x5EED1234 + scale
        h = C.c_void_p()
        check(self.lib.fgpu_mat_rmat(self._h, C.byref(h), scale, edge_factor, seed, a16, b16, c16))
        return Mat(self, h)


class Mat:
    """fgpu_mat: immutable device snapshot of one matrix layer."""

    def __init__(self, ctx: Context, h):
        self.ctx, self._h = ctx, h

    @property
    def handle(self):
        return self._h

    def free(self):
        if self._h:
            self.ctx.lib.fgpu_mat_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass

    def _get(self, fn):
        v = C.c_uint64()
        check(fn(self._h, C.byref(v)))
        return v.value

    @property
    def nrows(self):
        return self._get(self.ctx.lib.fgpu_mat_nrows)

    @property
    def ncols(self):
        return self._get(self.ctx.lib.fgpu_mat_ncols)

    @property
    def nvals(self):
        return self._get(self.ctx.lib.fgpu_mat_nvals)

    def export_csr(self):
        lib = self.ctx.lib
        rp, ci, vv = u64p(), u64p(), u64p()
        nnz = C.c_uint64()
        check(lib.fgpu_mat_export_csr(self.ctx._h, self._h, C.byref(rp), C.byref(ci), C.byref(vv), C.byref(nnz)))
        nrows = self.nrows
        rowptr = self.ctx._take(rp, nrows + 1)
        colidx = self.ctx._take(ci, max(nnz.value, 1))[: nnz.value]
        vals = self.ctx._take(vv, nnz.value) if vv else None
        return rowptr, colidx, vals

    def extract(self, min_row=0, max_row=2**64 - 1):
        lib = self.ctx.lib
        r, c, v = u64p(), u64p(), u64p()
        n = C.c_uint64()
        check(lib.fgpu_mat_extract(self.ctx._h, self._h, min_row, max_row, C.byref(r), C.byref(c), C.byref(v),
                                   C.byref(n)))
        rows = self.ctx._take(r, n.value)
        cols = self.ctx._take(c, n.value)
        vals = self.ctx._take(v, n.value) if v else None
        return rows, cols, vals

    def transpose(self) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_transpose(self.ctx._h, C.byref(h), self._h))
        return Mat(self.ctx, h)

    def balanced_splits(self, nparts: int):
        """nnz-balanced boundaries of the destination vertices (columns) for `nparts` column slabs (multiples of 4096)."""
        out = np.zeros(nparts + 1, dtype=U64)
        check(self.ctx.lib.fgpu_mat_balanced_splits(self.ctx._h, self._h, nparts, _p(out)))
        return out

    def sample(self, seed: int, denom: int) -> "Mat":
        """Entries (r, c) with mix64(seed ^ mix64(r << 32 | c)) % denom == 0 (bench / test data)."""
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_sample(self.ctx._h, C.byref(h), self._h, C.c_uint64(seed), C.c_uint32(denom)))
        return Mat(self.ctx, h)

    def probe(self, rows, cols, want_vals=False):
        rows, cols = _u64(rows), _u64(cols)
        present = np.zeros(len(rows), dtype=np.uint8)
        vals = np.zeros(len(rows), dtype=U64) if want_vals else None
        check(self.ctx.lib.fgpu_mat_probe(self.ctx._h, self._h, _p(rows), _p(cols), len(rows), _p(present, u8p),
                                          _p(vals)))
        return (present, vals) if want_vals else present

    def merge(self, dp: "Mat | None", dm: "Mat | None", dm_masks_dp=False) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_merge(self.ctx._h, C.byref(h), self._h, dp._h if dp else None,
                                          dm._h if dm else None, 1 if dm_masks_dp else 0))
        return Mat(self.ctx, h)

    def merge_pattern(self, dp: "Mat | None", dm: "Mat | None", dm_masks_dp=False) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_merge_pattern(self.ctx._h, C.byref(h), self._h, dp._h if dp else None,
                                                  dm._h if dm else None, 1 if dm_masks_dp else 0))
        return Mat(self.ctx, h)

    def resize(self, nrows, ncols) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_resize(self.ctx._h, C.byref(h), self._h, nrows, ncols))
        return Mat(self.ctx, h)

    def intersect(self, b: "Mat") -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_intersect(self.ctx._h, C.byref(h), self._h, b._h))
        return Mat(self.ctx, h)

    def intersect_nvals(self, b: "Mat") -> int:
        v = C.c_uint64()
        check(self.ctx.lib.fgpu_mat_intersect_nvals(self.ctx._h, self._h, b._h, C.byref(v)))
        return v.value

    def mxm(self, b: "Mat") -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mxm(self.ctx._h, C.byref(h), self._h, b._h))
        return Mat(self.ctx, h)

    def delta_lmxm(self, m: "Mat", dp: "Mat | None", dm: "Mat | None") -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_delta_lmxm(self.ctx._h, C.byref(h), self._h, m._h, dp._h if dp else None,
                                           dm._h if dm else None))
        return Mat(self.ctx, h)

    def build_tiles(self, tile_bits=0, vec=0, k=0):
        check(self.ctx.lib.fgpu_mat_build_tiles(self.ctx._h, self._h, tile_bits, vec, k))
        return self.tiles_info()

    def tiles_info(self):
        s = np.zeros(8, dtype=U64)
        check(self.ctx.lib.fgpu_mat_tiles_info(self._h, _p(s)))
        keys = ["tile_bits", "tiles", "groups", "items", "entries", "vec", "k", "bytes"]
        return dict(zip(keys, (int(x) for x in s)))

    def row_degrees(self, out_dev_ptr: int):
        """deg[r] = stored entries of row r, written to DEVICE memory (uint32[nrows], e.g. a torch int32 tensor)."""
        check(self.ctx.lib.fgpu_mat_row_degrees(self.ctx._h, self._h, C.c_void_p(out_dev_ptr)))

    def col_slab(self, lo, hi) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_col_slab(self.ctx._h, C.byref(h), self._h, lo, hi))
        return Mat(self.ctx, h)

    def row_slab(self, lo, hi) -> "Mat":
        h = C.c_void_p()
        check(self.ctx.lib.fgpu_mat_row_slab(self.ctx._h, C.byref(h), self._h, lo, hi))
        return Mat(self.ctx, h)


def _hop_arrays(mats):
    arr = (C.c_void_p * len(mats))()
    for i, m in enumerate(mats):
        arr[i] = m._h if m is not None else None
    return arr


def expand(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None):
    """fgpu_expand: the device core of CondTraverseOp::expand_batch.  Returns (rowptr, dest, flops)."""
    src = _u64(src_ids)
    nh = len(m)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    rp, ci = u64p(), u64p()
    nnz, flops = C.c_uint64(), C.c_uint64()
    check(ctx.lib.fgpu_expand(ctx._h, _p(src), len(src), am, adp, adm, nh, _p(lab), C.byref(rp), C.byref(ci),
                              C.byref(nnz), C.byref(flops)))
    rowptr = ctx._take(rp, len(src) + 1)
    dest = ctx._take(ci, max(nnz.value, 1))[: nnz.value]
    return rowptr, dest, flops.value


def expand32(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None):
    """fgpu_expand32: the same result in the device's own 32-bit form — (rowptr uint32[nsrc + 1], dest uint32[nnz], flops):
    two DMAs of the arrays as they lie, half the PCIe bytes of fgpu_expand."""
    src = _u64(src_ids)
    nh = len(m)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    u32p = C.POINTER(C.c_uint32)
    rp, ci = u32p(), u32p()
    nnz, flops = C.c_uint64(), C.c_uint64()
    check(ctx.lib.fgpu_expand32(ctx._h, _p(src), len(src), am, adp, adm, nh, _p(lab), C.byref(rp), C.byref(ci),
                                C.byref(nnz), C.byref(flops)))
    rowptr = ctx._take(rp, len(src) + 1, dtype=np.uint32)
    dest = ctx._take(ci, max(nnz.value, 1), dtype=np.uint32)[: nnz.value]
    return rowptr, dest, flops.value


class ExpandStream:
    """fgpu_expand_stream_*: the chain's result handed over in chunks of whole source rows while later chunks are still
    on the link.  Iterating yields (first_row, rowptr, dest) — views valid until the next step."""

    def __init__(self, ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None, chunk_rows=64, dest_bits=64):
        src = _u64(src_ids)
        am = _hop_arrays(m)
        adp = _hop_arrays(dp) if dp is not None else None
        adm = _hop_arrays(dm) if dm is not None else None
        lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
        self.ctx, self._h, self.dest_bits = ctx, C.c_void_p(), dest_bits
        nnz, flops = C.c_uint64(), C.c_uint64()
        check(ctx.lib.fgpu_expand_stream_open(ctx._h, _p(src), len(src), am, adp, adm, len(m), _p(lab), chunk_rows, dest_bits,
                                              C.byref(self._h), C.byref(nnz), C.byref(flops)))
        self.nnz, self.flops = nnz.value, flops.value

    def __iter__(self):
        return self

    def __next__(self):
        first, nrows = C.c_uint64(), C.c_uint64()
        rp, dest = u64p(), C.c_void_p()
        code = self.ctx.lib.fgpu_expand_stream_next(self._h, C.byref(first), C.byref(nrows), C.byref(rp), C.byref(dest))
        if code == _ffi.FGPU_NO_VALUE:
            raise StopIteration
        check(code)
        rowptr = np.ctypeslib.as_array(rp, shape=(nrows.value + 1,))
        n = int(rowptr[-1])
        dt = np.uint64 if self.dest_bits == 64 else np.uint32
        d = np.ctypeslib.as_array(C.cast(dest, C.POINTER(C.c_uint64 if self.dest_bits == 64 else C.c_uint32)), shape=(n,)) \
            if n else np.zeros(0, dtype=dt)
        return first.value, rowptr, d

    def close(self):
        if self._h:
            self.ctx.lib.fgpu_expand_stream_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h:
                self.close()
        except Exception:
            pass


def expand_mat(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None):
    """fgpu_expand_mat: the same chain with F left on the device as a matrix handle (cond_traverse.rs:602-608).
    Returns (Mat with len(src_ids) rows, flops)."""
    src = _u64(src_ids)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    h = C.c_void_p()
    flops = C.c_uint64()
    check(ctx.lib.fgpu_expand_mat(ctx._h, _p(src), len(src), am, adp, adm, len(m), _p(lab), C.byref(h), C.byref(flops)))
    return Mat(ctx, h), flops.value


def expand_count(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None, want_checksum=True):
    """fgpu_expand_count: (nnz, checksum, flops) of the k-hop result without materialising it on the host;
    want_checksum=False skips the per-entry hashing (count only: `RETURN count(c)`), checksum comes back 0."""
    src = _u64(src_ids)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    nnz, cs, flops = C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(ctx.lib.fgpu_expand_count(ctx._h, _p(src), len(src), am, adp, adm, len(m), _p(lab), C.byref(nnz),
                                    C.byref(cs) if want_checksum else None, C.byref(flops)))
    return nnz.value, cs.value, flops.value


def expand_trail_counts(ctx: Context, src_ids, m, dp=None, dm=None, weighted=False):
    """fgpu_expand_trail_counts: trails of exactly len(m) (1 or 2) hops per (source row, destination).
    Returns (rowptr, dest, count)."""
    src = _u64(src_ids)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    rp, ci, cv = u64p(), u64p(), u64p()
    nnz = C.c_uint64()
    check(ctx.lib.fgpu_expand_trail_counts(ctx._h, _p(src), len(src), am, adp, adm, len(m), 1 if weighted else 0,
                                           C.byref(rp), C.byref(ci), C.byref(cv), C.byref(nnz)))
    rowptr = ctx._take(rp, len(src) + 1)
    dest = ctx._take(ci, max(nnz.value, 1))[: nnz.value]
    count = ctx._take(cv, max(nnz.value, 1))[: nnz.value]
    return rowptr, dest, count


def expand_levels(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None):
    """fgpu_expand_levels: per-hop (nnz, checksum) of the chain and the DISTINCT union over the hops."""
    src = _u64(src_ids)
    nh = len(m)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    hn, hc = np.zeros(nh, dtype=U64), np.zeros(nh, dtype=U64)
    un, uc, fl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(ctx.lib.fgpu_expand_levels(ctx._h, _p(src), len(src), am, adp, adm, nh, _p(lab), _p(hn), _p(hc),
                                     C.byref(un), C.byref(uc), C.byref(fl)))
    return {"hop_nnz": hn.tolist(), "hop_checksum": hc.tolist(), "union_nnz": un.value,
            "union_checksum": uc.value, "flops": fl.value}


def vxm(ctx: Context, f_bits, mask_bits, A: Mat, At: Mat | None = None, direction=0):
    f = _u64(f_bits)
    mk = _u64(mask_bits) if mask_bits is not None else None
    w = np.zeros(len(f), dtype=U64)
    check(ctx.lib.fgpu_vxm(ctx._h, _p(w), _p(f), _p(mk), A._h, At._h if At else None, direction))
    return w


def bfs(ctx: Context, A: Mat, At: Mat | None, src: int, max_level: int = -1, want_parent: bool = True, level_out=None,
        parent_out=None):
    """fgpu_bfs.  level_out / parent_out: arrays to fill instead of fresh numpy ones — pass Context.host_array() blocks
    (pinned) and the library DMAs into them."""
    n = A.nrows
    level = level_out if level_out is not None else np.zeros(n, dtype=np.int32)
    parent = (parent_out if parent_out is not None else np.zeros(n, dtype=np.int64)) if want_parent else None
    edges = C.c_uint64()
    check(ctx.lib.fgpu_bfs(ctx._h, A._h, At._h if At else None, src, max_level, _p(level, i32p),
                           _p(parent, i64p), C.byref(edges)))
    return level, parent, edges.value


def pagerank(ctx: Context, A: Mat, At: Mat | None = None, active_bitmap=None, damping: float = 0.85,
             tol: float = 1e-4, itermax: int = 100):
    """fgpu_pagerank: LAGr_PageRank's numbers for algo.pageRank (FP32).  Returns (scores float32[n], iters)."""
    n = A.nrows
    out = np.zeros(n, dtype=np.float32)
    act = _u64(active_bitmap) if active_bitmap is not None else None
    it = C.c_int32()
    check(ctx.lib.fgpu_pagerank(ctx._h, A._h, At._h if At else None, _p(act), C.c_float(damping), C.c_float(tol),
                                C.c_int32(itermax), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(it)))
    return out, it.value


class BfsPlan:
    """fgpu_bfs_plan: resident BFS workspace (+ slab partition state for multi-rank runs)."""

    def __init__(self, ctx: Context, A: Mat, At: Mat | None, rank=0, nranks=1, splits=None):
        self.ctx, self.A, self.At = ctx, A, At
        self._h = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        if splits is None:
            check(ctx.lib.fgpu_bfs_plan_create(ctx._h, C.byref(self._h), A._h, At._h if At else None, rank, nranks))
        else:   # caller-chosen (nnz-balanced) slab boundaries
            sp = _u64(splits)
            assert len(sp) == nranks + 1
            check(ctx.lib.fgpu_bfs_plan_create_slab(ctx._h, C.byref(self._h), A._h, At._h if At else None, rank,
                                                    nranks, _p(sp)))
        self.n = A.nrows

    def dist_times(self):
        """(level-kernel ms, collective ms, level launches) of this rank's last bfs_dist_run (HIP-event sums)."""
        lm, cm, nl = C.c_double(), C.c_double(), C.c_uint64()
        check(self.ctx.lib.fgpu_bfs_dist_times(self._h, C.byref(lm), C.byref(cm), C.byref(nl)))
        return lm.value, cm.value, nl.value

    def free(self):
        if self._h:
            self.ctx.lib.fgpu_bfs_plan_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h:
                self.free()
        except Exception:
            pass

    def tune(self, alpha=0.0, beta=0.0, force_direction=0):
        check(self.ctx.lib.fgpu_bfs_plan_tune(self._h, alpha, beta, force_direction))

    def run(self, src, max_level=-1, want_parent=False):
        check(self.ctx.lib.fgpu_bfs_run(self._h, src, max_level, 1 if want_parent else 0))

    def run_async(self, src, max_level=-1, want_parent=False, levels=0):
        check(self.ctx.lib.fgpu_bfs_run_async(self._h, src, max_level, 1 if want_parent else 0, levels))

    def wait(self):
        check(self.ctx.lib.fgpu_bfs_wait(self._h))

    def fetch(self, want_parent=False):
        level = np.zeros(self.n, dtype=np.int32)
        parent = np.zeros(self.n, dtype=np.int64) if want_parent else None
        check(self.ctx.lib.fgpu_bfs_fetch(self._h, _p(level, i32p), _p(parent, i64p)))
        return level, parent

    def stats(self):
        s = np.zeros(8, dtype=U64)
        check(self.ctx.lib.fgpu_bfs_stats(self._h, _p(s)))
        keys = ["levels", "reached", "edges_traversed", "push_levels", "pull_levels", "scanned_push",
                "scanned_pull", "last_frontier"]
        return dict(zip(keys, (int(x) for x in s)))

    def profile(self, enable=True):
        check(self.ctx.lib.fgpu_bfs_plan_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        cap = 8
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_uint64 * cap)()
        ab = (C.c_uint64 * cap)()
        n = C.c_int()
        check(self.ctx.lib.fgpu_bfs_plan_profile_read(self._h, names, ms, launches, ab, cap, C.byref(n)))
        return [{"kernel": names[i].decode(), "ms": ms[i], "launches": launches[i], "alg_bytes": ab[i]}
                for i in range(n.value)]

    # ---- multi-rank stepping ----
    def part_buffers(self):
        lw, gw = C.c_void_p(), C.c_void_p()
        wpr = C.c_uint64()
        check(self.ctx.lib.fgpu_bfs_part_buffers(self._h, C.byref(lw), C.byref(gw), C.byref(wpr)))
        return lw.value, gw.value, wpr.value

    def part_set_buffers(self, local_ptr: int, global_ptr: int):
        check(self.ctx.lib.fgpu_bfs_part_set_buffers(self._h, C.c_void_p(local_ptr), C.c_void_p(global_ptr)))

    def part_begin(self, src, max_level=-1):
        check(self.ctx.lib.fgpu_bfs_part_begin(self._h, src, max_level))

    def part_step(self):
        check(self.ctx.lib.fgpu_bfs_part_step(self._h))

    def part_commit(self):
        check(self.ctx.lib.fgpu_bfs_part_commit(self._h))

    # ---- fused slab path (multi-rank v2: one kernel + one all-gather per level) ----
    def slab_set_buffers(self, send0_ptr: int, send1_ptr: int, global_ptr: int):
        check(self.ctx.lib.fgpu_bfs_slab_set_buffers(self._h, C.c_void_p(send0_ptr), C.c_void_p(send1_ptr),
                                                     C.c_void_p(global_ptr)))

    def slab_set_degrees(self, deg_ptr: int | None):
        check(self.ctx.lib.fgpu_bfs_slab_set_degrees(self._h, C.c_void_p(deg_ptr or 0)))

    def slab_begin(self, src, max_level=-1, want_parent=False):
        check(self.ctx.lib.fgpu_bfs_slab_begin(self._h, src, max_level, 1 if want_parent else 0))

    def slab_level(self) -> int:
        idx = C.c_int()
        check(self.ctx.lib.fgpu_bfs_slab_level(self._h, C.byref(idx)))
        return idx.value

    def part_done(self):
        d, l = C.c_int32(), C.c_int32()
        check(self.ctx.lib.fgpu_bfs_part_done(self._h, C.byref(d), C.byref(l)))
        return bool(d.value), l.value


def comm_init_all(ctxs):
    """fgpu_comm_init_all: one communicator over the contexts of ONE process (rank i = ctxs[i]) — the single-process
    gang form of the multi-GPU BFS (ncclCommInitAll inside libfgpu.so)."""
    ctxs = list(ctxs)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    check(ctxs[0].lib.fgpu_comm_init_all(arr, len(ctxs)))


def bfs_dist_run(plans, src: int, max_level: int = -1, want_parent: bool = False):
    """fgpu_bfs_dist_run: one whole search over a column-slab partition, level loop and frontier exchange inside the
    library.  `plans` = this process' ranks: one BfsPlan (one process per GPU, RCCL communicator on its context) or
    the plans of every rank in rank order (one process drives them all)."""
    plans = list(plans)
    arr = (C.c_void_p * len(plans))(*[p._h for p in plans])
    check(plans[0].ctx.lib.fgpu_bfs_dist_run(arr, len(plans), C.c_uint64(src), C.c_int64(max_level),
                                             1 if want_parent else 0))


def bench_spmv(ctx: Context, A: Mat, which=0, iters=20):
    ms = C.c_double()
    ab = C.c_uint64()
    check(ctx.lib.fgpu_bench_spmv(ctx._h, A._h, which, iters, C.byref(ms), C.byref(ab)))
    return ms.value, ab.value


def expand_pairs(ctx: Context, src_ids, m, dp=None, dm=None, dst_label_bitmap=None, pinned_dest=None, row_bits=16):
    """fgpu_expand_pairs: the (active_row, dest) columns CondTraverseOp::expand_batch hands on, built on the device;
    pinned_dest[i] = a pre-bound destination of source row i, or 2**64 - 1.  Returns (rows, dest, flops) as numpy arrays."""
    src = _u64(src_ids)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    pin = _u64(pinned_dest) if pinned_dest is not None else None
    prow, pdest = C.c_void_p(), u64p()
    n, flops = C.c_uint64(), C.c_uint64()
    check(ctx.lib.fgpu_expand_pairs(ctx._h, _p(src), len(src), am, adp, adm, len(m), _p(lab), _p(pin), int(row_bits),
                                    C.byref(prow), C.byref(pdest), C.byref(n), C.byref(flops)))
    k = n.value
    if k == 0:
        return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64), flops.value
    rt = np.uint16 if row_bits == 16 else np.uint32
    rows = np.ctypeslib.as_array(C.cast(prow, C.POINTER(C.c_uint16 if row_bits == 16 else C.c_uint32)), shape=(k,)).astype(np.uint64)
    dest = np.ctypeslib.as_array(pdest, shape=(k,)).copy()
    ctx.lib.fgpu_free(ctx._h, prow)
    ctx.lib.fgpu_free(ctx._h, pdest)
    return rows, dest, flops.value


def expand_probe(ctx: Context, src_ids, dst_ids, m, dp=None, dm=None, dst_label_bitmap=None):
    """fgpu_expand_probe: present[i] = dst_ids[i] is reached from src_ids[i] by the chain (every row of the batch has a
    pre-bound destination); returns (present as a bool array, flops of the hops that ran)."""
    src, dst = _u64(src_ids), _u64(dst_ids)
    assert len(src) == len(dst)
    am = _hop_arrays(m)
    adp = _hop_arrays(dp) if dp is not None else None
    adm = _hop_arrays(dm) if dm is not None else None
    lab = _u64(dst_label_bitmap) if dst_label_bitmap is not None else None
    out = np.zeros(max(len(src), 1), dtype=np.uint8)
    flops = C.c_uint64()
    check(ctx.lib.fgpu_expand_probe(ctx._h, _p(src), _p(dst), len(src), am, adp, adm, len(m), _p(lab),
                                    out.ctypes.data_as(u8p), C.byref(flops)))
    return out[:len(src)].astype(bool), flops.value
