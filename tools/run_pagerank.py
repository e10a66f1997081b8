"""Profiling target: a few fgpu_pagerank calls on RMAT-<scale> (rocprofv3 --kernel-trace --stats -- python tools/run_pagerank.py 22 [parts])."""
import sys
sys.path.insert(0, ".")
from falkordb_amd import engine
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = engine.Context(0)
if len(sys.argv) > 2:
    ctx.set_option("pagerank_parts", int(sys.argv[2]))
A = ctx.mat_rmat(scale)
At = A.transpose()
for _ in range(3):
    s, it = engine.pagerank(ctx, A, At, None, 0.85, 0.0, 10)
print(it, float(s.sum()))
