"""Which of the bench's BFS roots take the propagation-blocking path: per root ms, levels, push/pull, pb levels.
usage: python tools/experiments/pb_roots.py [scale] [nroots]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from falkordb_amd import engine
from bench import pick_roots
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = engine.Context(0)
A = ctx.mat_rmat(scale); At = A.transpose()
roots = pick_roots(A, 64)[:nr]
plan = engine.BfsPlan(ctx, A, At)
plan.run(int(roots[0]))
for i, r in enumerate(roots):
    ts = []
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter(); plan.run(int(r)); ts.append(time.perf_counter() - t0)
    st = plan.stats()
    print(i, int(r), "ms %.3f" % (min(ts) * 1e3), "levels", st["levels"], "push", st["push_levels"], "pull", st["pull_levels"],
          "scanned_push", st["scanned_push"], "pb", ctx.get_option("bfs_pb_last_levels"), flush=True)
