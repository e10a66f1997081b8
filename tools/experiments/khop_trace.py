"""16 clean RMAT-22 3-hop batches for a rocprofv3 --kernel-trace run (tools/trace_batches.py reads the trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from falkordb_amd import engine
ctx = engine.Context(0)
A = ctx.mat_rmat(22, 16, 0x5EED1234 + 22)
srcs = bench.p_label_sources(A.nrows)
batches = [srcs[j * 1024:(j + 1) * 1024] for j in range(16)]
for b in batches[:3]:
    engine.expand_count(ctx, b, [A] * 3)
ctx.sync()
time.sleep(0.01)
for b in batches:
    engine.expand_count(ctx, b, [A] * 3)
    time.sleep(0.001)
ctx.sync()
