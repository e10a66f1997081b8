"""Phase stamps of the fused level kernel in SLAB mode (one rank, fgpu_bfs_dist_run): library built with -DFGPU_BFS_STAMPS."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from falkordb_amd import engine
from bench import pick_roots
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = engine.Context(0)
A = ctx.mat_rmat(scale); At = A.transpose()
roots = pick_roots(A, 64)
plan = engine.BfsPlan(ctx, A, At, 0, 1, splits=A.balanced_splits(1))
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
lib = ctx.lib
lib.fgpu_debug_bfs_stamps.argtypes = [C.c_void_p]
for ri in (17,):
    root = roots[ri]
    engine.bfs_dist_run([plan], root, -1, False)
    full = plan.stats()
    for L in range(1, full["levels"] + 1):
        buf.zero_()
        lib.fgpu_debug_bfs_stamps(C.c_void_p(buf.data_ptr()))
        engine.bfs_dist_run([plan], root, L, False)
        lib.fgpu_debug_bfs_stamps(None)
        torch.cuda.synchronize()
        b = buf.cpu().numpy().reshape(-1, 8)
        b = b[b[:, 0] > 0]
        t0 = b[:, 0].min()
        line = f"level {L}: wgs {len(b)}"
        for k, name in [(1, "items"), (2, "hubs"), (3, "work"), (4, "ticket"), (5, "end")]:
            x = (b[:, k] - t0) / 100.0
            x = x[(b[:, k] > 0) & (x > 0)]
            if len(x): line += f" | {name} p50 {np.median(x):5.1f} max {x.max():5.1f}"
        print(line)
