"""The emitting form of the chain (fgpu_expand_mat / fgpu_expand: what CondTraverse walks, cond_traverse.rs:644-751) with the
bit state turned into rows by the ballot transpose (expand_emit_sort = 0, rounds 3-5), by pairs + the stable sort (2) and by
the density rule (1, the default):
ms per batch on the device, into host arrays, the kernel table, and the two results compared array for array.
usage: python tools/emit_ab.py [scale=24] [hops=2] [rows=1024] [batches=6]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from falkordb_amd import engine  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 6
ctx = engine.Context(0)
A = ctx.mat_rmat(scale, 16, 0x5EED1234 + scale)
srcs = bench.p_label_sources(A.nrows)
bl = [srcs[j * 1024:j * 1024 + rows] for j in range(nb)]
ref = None
for mode in (0, 2, 1):
    ctx.set_option("expand_emit_sort", mode)
    for b in bl[:2]:
        m_, _ = engine.expand_mat(ctx, b, [A] * hops)
        m_.free()
    ctx.sync()
    t = time.perf_counter()
    nn = 0
    for b in bl:
        m_, f = engine.expand_mat(ctx, b, [A] * hops)
        nn += m_.nvals
        m_.free()
    ctx.sync()
    dt = (time.perf_counter() - t) / nb
    th = []
    for rep in range(2):
        for b in bl[:4]:
            t = time.perf_counter()
            r_ = engine.expand(ctx, b, [A] * hops)
            th.append(time.perf_counter() - t)
            if b is bl[0] and rep == 1:
                got = (r_[0].copy(), r_[1].copy())
            del r_
    ctx.prof_enable(True)
    for b in bl:
        m_, f = engine.expand_mat(ctx, b, [A] * hops)
        m_.free()
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    same = None
    if ref is None:
        ref = got
    else:
        same = bool(np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]))
    print(json.dumps({"emit_sort": mode, "scale": scale, "hops": hops, "rows": rows, "ms_device": round(dt * 1e3, 3),
                      "ms_host_arrays": round(sorted(th[len(th) // 2:])[len(th) // 4] * 1e3, 3), "entries": int(nn // nb),
                      "same_as_first": same,
                      "kernels": {k["kernel"]: round(k["ms"] / k["launches"] * 1e3, 1) for k in sorted(prof, key=lambda k: -k["ms"])[:9]}}),
          flush=True)
