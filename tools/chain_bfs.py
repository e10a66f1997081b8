"""High-diameter case: BFS down a path graph (every level holds one vertex), bfs_tiny off / on."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from falkordb_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = engine.Context(0)
rows = np.arange(n - 1, dtype=np.uint64); cols = rows + np.uint64(1)
A = ctx.mat_from_coo(n, n, rows, cols); At = A.transpose()
for tiny in (0, 1):
    ctx.set_option("bfs_tiny", tiny)
    plan = engine.BfsPlan(ctx, A, At)
    plan.run(0); plan.run(0)
    ctx.sync(); t0 = time.perf_counter(); plan.run(0); dt = time.perf_counter() - t0
    lv, _ = plan.fetch()
    assert (lv == np.arange(n)).all()
    print(f"bfs_tiny={tiny}: {n - 1} levels in {dt * 1e3:.2f} ms = {dt / (n - 1) * 1e6:.2f} us per level")
