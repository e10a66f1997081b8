"""One RMAT-<scale> full-pass boolean SpMV workload for rocprofv3 (tools/prof_spmv.sh)."""
import sys
sys.path.insert(0, ".")
from falkordb_amd import engine

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = engine.Context(0)
A = ctx.mat_rmat(scale)
At = A.transpose()
info = At.build_tiles()
ms, ab = engine.bench_spmv(ctx, At, 2, iters)
print(f"tiled full pass: {ms*1e3:.2f} us/launch, alg bytes {ab}, {ab/ms/1e6:.0f} GB/s, layout {info}")
ms, ab = engine.bench_spmv(ctx, At, 0, 5)
print(f"csr pull full pass: {ms*1e3:.2f} us/launch, {ab/ms/1e6:.0f} GB/s")
