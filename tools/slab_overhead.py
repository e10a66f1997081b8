"""Per-BFS cost of the multi-rank (column-slab) level loop driven from Python on ONE GPU (nranks = 1, no
collective): what the step / commit split and the host loop cost next to the fused single-rank path."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from falkordb_amd import engine, dist as fdist

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = engine.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
A = ctx.mat_rmat(scale); At = A.transpose()
rows, _, _ = A.extract(0, 4095)
roots = np.unique(rows)[:64].tolist()
back = fdist.HipSlabBackend(ctx, A, At, 0, 1, dev, "fused")
old = fdist.HipSlabBackend(ctx, A, At, 0, 1, dev, "stepped")
fused = engine.BfsPlan(ctx, A, At)
for name, fn in (("slab v2 (python: 1 level kernel + gather per level)", lambda r: back.run(int(r))),
                 ("slab v1 (python: step, gather, commit per level)", lambda r: old.run(int(r))),
                 ("single-rank fused plan.run", lambda r: fused.run(int(r), -1, False))):
    for r in roots[:8]:
        fn(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in roots:
        fn(r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(roots)
    print(f"{name}: {dt*1e3:.3f} ms per BFS")
st = back.plan.stats()
print("levels", st["levels"], "push", st["push_levels"], "pull", st["pull_levels"])
