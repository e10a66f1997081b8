"""16 DIRTY RMAT-22 3-hop batches (bench.py khop_inputs' layers) for a rocprofv3 --kernel-trace run (tools/trace_batches.py)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from falkordb_amd import engine
ctx = engine.Context(0)
A, dp, dm, _ = bench.khop_inputs(ctx, 22, 16)
srcs = bench.p_label_sources(A.nrows)
batches = [srcs[j * 1024:(j + 1) * 1024] for j in range(16)]
lay = ([A] * 3, [dp] * 3, [dm] * 3)
for b in batches[:3]:
    engine.expand_count(ctx, b, *lay)
ctx.sync()
time.sleep(0.01)
for b in batches:
    engine.expand_count(ctx, b, *lay)
    time.sleep(0.001)
ctx.sync()
