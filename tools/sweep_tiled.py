"""Timing sweep of the LDS-tiled full-pass boolean SpMV on RMAT-<scale> (run on the GPU box)."""
import sys, json, itertools
sys.path.insert(0, ".")
from falkordb_amd import engine

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = engine.Context(0)
A = ctx.mat_rmat(scale)
At = A.transpose()
ms, ab = engine.bench_spmv(ctx, At, 0, 10)
print(f"csr pull: {ms*1e3:.1f} us  {ab/ms/1e6:.0f} GB/s", flush=True)
configs = [(20, 4, 1), (20, 4, 2), (20, 2, 1), (20, 2, 2)]
if len(sys.argv) > 2:
    configs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[2:]]
for tb, vec, k in configs:
    info = At.build_tiles(tb, vec, k)
    for threads in (1024,):
        ctx.set_option("tiled_threads", threads)
        for nt in (0, 1):
            ctx.set_option("tiled_nt", nt)
            for u in (2, 4, 8):
                ctx.set_option("tiled_u", u)
                ms, ab = engine.bench_spmv(ctx, At, 2, 30)
                print(f"tb={tb} vec={vec} k={k} thr={threads} nt={nt} u={u}: {ms*1e3:.1f} us {ab/ms/1e6:.0f} GB/s "
                      f"items={info['items']} entries={info['entries']} bytes={info['bytes']/1e6:.0f}MB", flush=True)
