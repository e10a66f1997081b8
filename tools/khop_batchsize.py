#!/usr/bin/env python3
"""Does a narrower bit state (fewer source rows per batch) make the dense last hop cheaper per source?  Same RMAT-24 inputs
as bench.py; per batch size: ms per batch, ms per 1024 sources, the top kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from falkordb_amd import engine
ctx = engine.Context(0)
A, dp, dm, _ = bench.khop_inputs(ctx, 24, 16)
srcs = bench.p_label_sources(A.nrows)
for B in (1024, 512, 256, 128, 64):
    nb = 4096 // B
    batches = [srcs[i * B:(i + 1) * B] for i in range(nb)]
    for b in batches[:2]:
        engine.expand_count(ctx, b, [A] * 3)
    ctx.sync()
    t0 = time.perf_counter()
    tot = 0
    for b in batches:
        n, c, f = engine.expand_count(ctx, b, [A] * 3)
        tot += f
    dt = time.perf_counter() - t0
    ctx.prof_enable(True)
    for b in batches:
        engine.expand_count(ctx, b, [A] * 3)
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    top = sorted(prof, key=lambda k: -k["ms"])[:3]
    print(json.dumps({"batch_rows": B, "ms_per_batch": round(dt / nb * 1e3, 3), "ms_per_1024_sources": round(dt / nb * 1e3 * 1024 / B, 3),
                      "TEPS": round(tot / dt / 1e9, 1), "top": [(k["kernel"], round(k["ms"] / k["launches"], 3)) for k in top]}), flush=True)
