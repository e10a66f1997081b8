"""ctypes binding of libfgpu.so's C ABI (include/fgpu.h).

There is no CPU fallback: importing works without a GPU (so the ABI can be inspected and the
export list checked), but `Context()` fails loudly when no HIP device is present, and a missing
library raises instead of degrading.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGPU_LIB") or os.path.join(_HERE, "lib", "libfgpu.so")

FGPU_OK = 0
FGPU_NO_VALUE = 1
FGPU_NULL_POINTER = -2
FGPU_INVALID = -3
FGPU_DIM_MISMATCH = -6
FGPU_OOM = -102
FGPU_OUT_OF_BOUNDS = -105
FGPU_DEVICE = -7002

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); the single source of truth checked against include/fgpu.h by
# tests/test_abi.py
SIGNATURES = {
    "fgpu_init": (C.c_int32, [vpp, C.c_int, vp, vp]),
    "fgpu_finalize": (C.c_int32, [vp]),
    "fgpu_last_error": (C.c_char_p, []),
    "fgpu_free": (None, [vp, vp]),
    "fgpu_host_alloc": (C.c_int32, [vp, C.c_uint64, vpp]),
    "fgpu_expand_stream_open": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, C.c_uint64, C.c_int, vpp,
                                            u64p, u64p]),
    "fgpu_expand_stream_next": (C.c_int32, [vp, u64p, u64p, C.POINTER(u64p), vpp]),
    "fgpu_expand_stream_close": (C.c_int32, [vp]),
    "fgpu_set_stream": (C.c_int32, [vp, vp]),
    "fgpu_sync": (C.c_int32, [vp]),
    "fgpu_set_option": (C.c_int32, [vp, C.c_char_p, C.c_int64]),
    "fgpu_get_option": (C.c_int32, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "fgpu_mat_build_tiles": (C.c_int32, [vp, vp, C.c_int, C.c_int, C.c_int]),
    "fgpu_mat_tiles_info": (C.c_int32, [vp, u64p]),
    "fgpu_device_info": (C.c_int32, [vp, C.c_char_p, i32p, i32p, i64p, i64p]),
    "fgpu_device_bytes": (C.c_int32, [vp, u64p, u64p]),
    "fgpu_mat_new": (C.c_int32, [vp, vpp, C.c_uint64, C.c_uint64]),
    "fgpu_mat_from_coo": (C.c_int32, [vp, vpp, C.c_uint64, C.c_uint64, u64p, u64p, u64p, C.c_uint64]),
    "fgpu_mat_from_csr": (C.c_int32, [vp, vpp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_int, vp, C.c_int,
                                      u64p, u64p, C.c_uint64]),
    "fgpu_mat_rmat": (C.c_int32, [vp, vpp, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "fgpu_mat_free": (C.c_int32, [vp]),
    "fgpu_mat_nrows": (C.c_int32, [vp, u64p]),
    "fgpu_mat_ncols": (C.c_int32, [vp, u64p]),
    "fgpu_mat_nvals": (C.c_int32, [vp, u64p]),
    "fgpu_mat_has_values": (C.c_int32, [vp, i32p]),
    "fgpu_mat_export_csr": (C.c_int32, [vp, vp, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u64p), u64p]),
    "fgpu_mat_extract": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(u64p), C.POINTER(u64p),
                                     C.POINTER(u64p), u64p]),
    "fgpu_mat_transpose": (C.c_int32, [vp, vpp, vp]),
    "fgpu_mat_probe": (C.c_int32, [vp, vp, u64p, u64p, C.c_uint64, u8p, u64p]),
    "fgpu_mat_merge": (C.c_int32, [vp, vpp, vp, vp, vp, C.c_int]),
    "fgpu_mat_merge_pattern": (C.c_int32, [vp, vpp, vp, vp, vp, C.c_int]),
    "fgpu_mat_resize": (C.c_int32, [vp, vpp, vp, C.c_uint64, C.c_uint64]),
    "fgpu_mat_intersect": (C.c_int32, [vp, vpp, vp, vp]),
    "fgpu_mat_intersect_nvals": (C.c_int32, [vp, vp, vp, u64p]),
    "fgpu_mxm": (C.c_int32, [vp, vpp, vp, vp]),
    "fgpu_delta_lmxm": (C.c_int32, [vp, vpp, vp, vp, vp, vp]),
    "fgpu_expand": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, C.POINTER(u64p),
                                C.POINTER(u64p), u64p, u64p]),
    "fgpu_expand32": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, C.POINTER(u32p),
                                  C.POINTER(u32p), u64p, u64p]),
    "fgpu_expand_mat": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, vpp, u64p]),
    "fgpu_expand_probe": (C.c_int32, [vp, u64p, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, u8p, u64p]),
    "fgpu_expand_pairs": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, u64p, C.c_int, vpp, C.POINTER(u64p), u64p, u64p]),
    "fgpu_expand_pairs32": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, u64p, C.c_int, vpp, C.POINTER(u32p), u64p, u64p]),
    "fgpu_expand_levels": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, u64p, u64p, u64p, u64p, u64p]),
    "fgpu_expand_trail_counts": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, C.c_int, C.POINTER(u64p),
                                             C.POINTER(u64p), C.POINTER(u64p), u64p]),
    "fgpu_expand_count": (C.c_int32, [vp, u64p, C.c_uint64, vpp, vpp, vpp, C.c_int, u64p, u64p, u64p, u64p]),
    "fgpu_vxm": (C.c_int32, [vp, u64p, u64p, u64p, vp, vp, C.c_int]),
    "fgpu_bfs": (C.c_int32, [vp, vp, vp, C.c_uint64, C.c_int64, i32p, i64p, u64p]),
    "fgpu_pagerank": (C.c_int32, [vp, vp, vp, u64p, C.c_float, C.c_float, C.c_int32, C.POINTER(C.c_float), i32p]),
    "fgpu_pagerank_status": (C.c_int32, [vp, vp, vp, u64p, C.c_float, C.c_float, C.c_int32, C.POINTER(C.c_float), i32p, i32p]),
    "fgpu_bfs_plan_create": (C.c_int32, [vp, vpp, vp, vp, C.c_int, C.c_int]),
    "fgpu_bfs_plan_free": (C.c_int32, [vp]),
    "fgpu_bfs_plan_tune": (C.c_int32, [vp, C.c_double, C.c_double, C.c_int]),
    "fgpu_bfs_run": (C.c_int32, [vp, C.c_uint64, C.c_int64, C.c_int]),
    "fgpu_bfs_run_async": (C.c_int32, [vp, C.c_uint64, C.c_int64, C.c_int, C.c_int]),
    "fgpu_bfs_wait": (C.c_int32, [vp]),
    "fgpu_bfs_fetch": (C.c_int32, [vp, i32p, i64p]),
    "fgpu_bfs_stats": (C.c_int32, [vp, u64p]),
    "fgpu_bfs_part_buffers": (C.c_int32, [vp, vpp, vpp, u64p]),
    "fgpu_bfs_part_set_buffers": (C.c_int32, [vp, vp, vp]),
    "fgpu_bfs_part_begin": (C.c_int32, [vp, C.c_uint64, C.c_int64]),
    "fgpu_bfs_part_step": (C.c_int32, [vp]),
    "fgpu_bfs_part_commit": (C.c_int32, [vp]),
    "fgpu_bfs_part_done": (C.c_int32, [vp, i32p, i32p]),
    "fgpu_bfs_slab_set_buffers": (C.c_int32, [vp, vp, vp, vp]),
    "fgpu_bfs_slab_set_degrees": (C.c_int32, [vp, vp]),
    "fgpu_bfs_slab_begin": (C.c_int32, [vp, C.c_uint64, C.c_int64, C.c_int]),
    "fgpu_bfs_slab_level": (C.c_int32, [vp, C.POINTER(C.c_int)]),
    "fgpu_mat_row_degrees": (C.c_int32, [vp, vp, vp]),
    "fgpu_mat_col_slab": (C.c_int32, [vp, vpp, vp, C.c_uint64, C.c_uint64]),
    "fgpu_mat_row_slab": (C.c_int32, [vp, vpp, vp, C.c_uint64, C.c_uint64]),
    "fgpu_bench_spmv": (C.c_int32, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_double), u64p]),
    "fgpu_comm_unique_id": (C.c_int32, [u8p]),
    "fgpu_comm_init_rank": (C.c_int32, [vp, C.c_int, C.c_int, u8p]),
    "fgpu_comm_init_all": (C.c_int32, [vpp, C.c_int]),
    "fgpu_comm_finalize": (C.c_int32, [vp]),
    "fgpu_comm_info": (C.c_int32, [vp, i32p, i32p]),
    "fgpu_mat_balanced_splits": (C.c_int32, [vp, vp, C.c_int, u64p]),
    "fgpu_splits_shift": (C.c_uint32, [C.c_uint64]),
    "fgpu_balanced_splits_from_hist": (C.c_int32, [u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, u64p]),
    "fgpu_slab_layout": (C.c_int32, [u64p, C.c_uint64, C.c_int, u64p, u64p, u64p, u64p]),
    "fgpu_bfs_plan_create_slab": (C.c_int32, [vp, vpp, vp, vp, C.c_int, C.c_int, u64p]),
    "fgpu_bfs_dist_run": (C.c_int32, [vpp, C.c_int, C.c_uint64, C.c_int64, C.c_int]),
    "fgpu_bfs_dist_times": (C.c_int32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p]),
    "fgpu_prof_enable": (C.c_int32, [vp, C.c_int]),
    "fgpu_prof_read": (C.c_int32, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), u64p, u64p, C.c_int,
                                   C.POINTER(C.c_int)]),
    "fgpu_mat_sample": (C.c_int32, [vp, vpp, vp, C.c_uint64, C.c_uint32]),
    "fgpu_bfs_plan_profile": (C.c_int32, [vp, C.c_int]),
    "fgpu_bfs_plan_profile_read": (C.c_int32, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), u64p, u64p,
                                               C.c_int, C.POINTER(C.c_int)]),
}

_lib = None


class FgpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fgpu error {code}: {msg}")
        self.code = code


def load():
    """Load libfgpu.so (building it first if the sources are newer / it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build_lib()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"libfgpu.so not found at {LIB_PATH}; run `python -m falkordb_amd.build`")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => the library does not export what fgpu.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int):
    if code != FGPU_OK:
        msg = load().fgpu_last_error()
        raise FgpuError(code, msg.decode("utf-8", "replace") if msg else "")
