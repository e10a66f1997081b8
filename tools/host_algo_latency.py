"""Procedure latency through the host layer on a committed graph: algo.BFS / algo.pageRank, first and repeated calls
(adjacency extract, transpose and the BFS acceleration indexes are shared between calls once the layers are clean)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from falkordb_amd import engine, host
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
hc = host.Context(0)
ctx = engine.Context(0)
A = ctx.mat_rmat(scale)
n = A.nrows
rp, ci, _ = A.export_csr()
r = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp).astype(np.int64))
g = host.Graph(hc, n)
t = g.add_type("KNOWS")
g.create_edges(t, r, ci, np.arange(len(ci), dtype=np.uint64))
g.commit()
src = int(np.argmax(np.diff(rp)))
for name, fn in (("algo.BFS", lambda: g.algo_bfs(src)), ("algo.pageRank", lambda: g.algo_pagerank(None, None))):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(((time.perf_counter() - t0) * 1e3, g.L.fh_last_op_ns() / 1e6))
    print(f"{name} scale {scale}: C++ operator ms " + " ".join(f"{c:.2f}" for _, c in ts) +
          "   (through the ctypes harness: " + " ".join(f"{x:.1f}" for x, _ in ts) + ")")
