#!/usr/bin/env python3
"""Full-pass boolean SpMV: the tiled (tiled.hip) and blocked (blocked.hip) layouts at one scale, warm and cold, with the
CSR pull beside them.   python tools/spmv_layouts.py 22 24 26"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from falkordb_amd import engine  # noqa: E402

ctx = engine.Context(0)
for scale in [int(x) for x in sys.argv[1:]] or [22]:
    A = ctx.mat_rmat(scale)
    At = A.transpose()
    A.free()
    rec = {"scale": scale, "edges": int(At.nvals)}
    for name, lay, u in (("tiled", 1, 4), ("blocked pf3 2wg", 2, 4), ("blocked pf4 1wg", 2, 8), ("blocked pf2 2wg", 2, 2),
                         ("blocked pf4 2wg (spills)", 2, 1)):
        ctx.set_option("tiled_layout", lay)
        ctx.set_option("blocked_variant", {4: 0, 2: 1, 8: 2, 1: 3}[u])
        info = At.build_tiles()
        ms, ab = engine.bench_spmv(ctx, At, which=2, iters=20)
        ms_c, _ = engine.bench_spmv(ctx, At, which=3, iters=8)
        rec[name] = {"warm_us": round(ms * 1e3, 1), "cold_us": round(ms_c * 1e3, 1), "frac_warm": round(ab / ms / 1e6 / 8000, 3),
                     "frac_cold": round(ab / ms_c / 1e6 / 8000, 3), "layout_bytes": info["bytes"], "entries": info["entries"],
                     "tile_bits": info["tile_bits"], "items_or_blocks": info["items"]}
    ctx.set_option("blocked_variant", 0)
    ms0, _ = engine.bench_spmv(ctx, At, which=0, iters=5)
    rec["csr_pull_us"] = round(ms0 * 1e3, 1)
    print(json.dumps(rec), flush=True)
    At.free()
