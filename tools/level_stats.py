"""Per-level frontier / work / time table of one BFS (run with max_level = 1..L and difference the plan
statistics; time = median wall time of the k-level search minus the (k-1)-level one).
usage: python tools/level_stats.py [scale] [root_index] [alpha]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from falkordb_amd import engine
from bench import pick_roots

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ri = int(sys.argv[2]) if len(sys.argv) > 2 else 0
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ctx = engine.Context(0)
A = ctx.mat_rmat(scale)
At = A.transpose()
root = pick_roots(A, 64)[ri]
plan = engine.BfsPlan(ctx, A, At)
if alpha:
    plan.tune(alpha=alpha)
plan.run(root)
full = plan.stats()
prev, tprev = None, 0.0
print("root", root, full)
for k in range(1, full["levels"] + 1):
    ts = []
    for _ in range(7):
        ctx.sync()
        t0 = time.perf_counter()
        plan.run(root, k)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    s = plan.stats()
    d = {x: s[x] - (prev[x] if prev else 0) for x in s}
    direction = "pull" if d["pull_levels"] else "push"
    print(f"level {k}: {direction}  new={d['reached']:>8}  m_f(out-deg of new)={d['edges_traversed']:>9}  "
          f"scanned={d['scanned_push'] + d['scanned_pull']:>9}  dt={1e6 * (t - tprev):7.1f} us  (cum {1e6 * t:7.1f})")
    prev, tprev = s, t
