import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from falkordb_amd import engine
ctx = engine.Context(0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
A = ctx.mat_rmat(scale); At = A.transpose(); A.free()
for lay in (2, 1):
    ctx.set_option("tiled_layout", lay)
    At.build_tiles()
    print(lay, engine.bench_spmv(ctx, At, which=2, iters=4))
