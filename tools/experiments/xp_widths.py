import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle
from falkordb_amd import engine
import test_gpu_scale as T
ctx = engine.Context(0)
A = ctx.mat_rmat(20)
a = T.host_csr(A)
n = a.nrows
allsrc = T.p_sources(n, 1024)
ctx.set_option("expand_mode", 2)
ctx.set_option("expand_xcd_min_mb", 0)
for nsrc in (64, 100, 128, 129, 160, 192, 200, 256, 300, 400, 448, 512, 640, 1024):
    src = allsrc[:nsrc]
    ctx.set_option("expand_xcd", 0)
    ref = engine.expand_count(ctx, src, [A] * 3)
    ctx.set_option("expand_xcd", 1)
    got = [engine.expand_count(ctx, src, [A] * 3) for _ in range(3)]
    print(nsrc, "w", (nsrc + 63) // 64, "OK" if all(g == ref for g in got) else ("MISMATCH", ref, got), flush=True)
