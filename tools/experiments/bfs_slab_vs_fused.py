#!/usr/bin/env python3
"""Same roots through the single-rank fused search and through the one-rank slab path: levels, directions, scanned edges."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from falkordb_amd import engine
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ctx = engine.Context(0)
A = ctx.mat_rmat(scale, 16, 0x5EED1234 + scale)
At = A.transpose()
roots = bench.pick_roots(A, 8)
fused = engine.BfsPlan(ctx, A, At)
splits = A.balanced_splits(1)
slab = engine.BfsPlan(ctx, A, At, 0, 1, splits=splits)
for r in roots[:6]:
    out = {}
    for name in ("fused", "slab"):
        for rep in range(2):
            ctx.sync()
            t = time.perf_counter()
            if name == "fused":
                fused.run(r, -1, False)
            else:
                engine.bfs_dist_run([slab], r, -1, False)
            ctx.sync()
            dt = time.perf_counter() - t
        st = (fused if name == "fused" else slab).stats()
        out[name] = dict(ms=round(dt * 1e3, 3), **{k: st[k] for k in ("levels", "push_levels", "pull_levels", "scanned_push", "scanned_pull", "reached")})
    print(json.dumps({"root": int(r), **out}), flush=True)
