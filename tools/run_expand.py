"""k-hop CondTraverse core (fgpu_expand_count) on RMAT-<scale>: sources = every 16th-hash vertex, batched."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from falkordb_amd import engine

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
hops = int(sys.argv[2]) if len(sys.argv) > 2 else 3
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nbatches = int(sys.argv[4]) if len(sys.argv) > 4 else 3
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = engine.Context(0)
ctx.set_option("expand_mode", mode)
t0 = time.time()
A = ctx.mat_rmat(scale)
ctx.sync()
n = A.nrows
print(f"rmat-{scale}: n={n} nnz={A.nvals} build {time.time()-t0:.2f}s  expand_mode={mode} batch={batch}", flush=True)
rng = np.random.default_rng(7)
for b in range(nbatches):
    src = rng.choice(n, batch, replace=False).astype(np.uint64)
    for h in range(1, hops + 1):
        ctx.sync()
        t0 = time.perf_counter()
        nnz, cs, flops = engine.expand_count(ctx, src, [A] * h)
        dt = time.perf_counter() - t0
        print(f"batch {b} hops {h}: nnz={nnz} flops={flops} {dt*1e3:.2f} ms  {flops/dt/1e9:.2f} GTEPS  "
              f"out {nnz/dt/1e9:.2f} Gnnz/s", flush=True)
