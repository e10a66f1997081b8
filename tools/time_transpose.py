"""Wall time + per-kernel HIP-event times of fgpu_mat_transpose / the R-MAT COO build (transpose.hip), every mode.
usage: python tools/time_transpose.py [scales...]"""
import sys
import time

sys.path.insert(0, ".")
from falkordb_amd import engine

ctx = engine.Context(0)
for scale in [int(x) for x in sys.argv[1:]] or [22, 24]:
    for mode in (3, 2, 1):
        ctx.set_option("transpose_mode", mode)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); A = ctx.mat_rmat(scale); ctx.sync(); ts.append(time.perf_counter() - t0)
            if _ < 2:
                A.free()
        tb = min(ts)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); T = A.transpose(); ctx.sync(); ts.append(time.perf_counter() - t0); T.free()
        n, nnz = A.nrows, A.nvals
        alg = 8 * nnz + 8 * (n + 1)     # read colidx + rowptr, write colidx' + rowptr'
        print({"scale": scale, "mode": {3: "counting, two levels", 2: "counting, LDS-staged levels", 1: "coo+sort"}[mode], "rmat_build_ms": round(tb * 1e3, 2),
               "transpose_ms": [round(t * 1e3, 2) for t in ts], "nnz": nnz,
               "transpose_GBps": round(alg / min(ts) / 1e9, 1), "frac_hbm": round(alg / min(ts) / 8e12, 4)}, flush=True)
        if mode != 1:
            ctx.prof_enable(True)
            T = A.transpose()
            for k in ctx.prof_read():
                print("   ", k["kernel"], round(k["ms"], 3), "ms")
            ctx.prof_enable(False)
            T.free()
        A.free()
