"""Debug aid: max_level searches vs the truncated full search, with stats cross-checks."""
import sys
sys.path.insert(0, ".")
import numpy as np
from falkordb_amd import engine
from bench import pick_roots
ctx = engine.Context(0)
A = ctx.mat_rmat(22); At = A.transpose()
rp, ci, _ = A.export_csr(); rp = rp.astype(np.int64); deg = np.diff(rp)
trp, tci, _ = At.export_csr(); trp = trp.astype(np.int64)
roots = pick_roots(A, 64)
plan = engine.BfsPlan(ctx, A, At)
if len(sys.argv) > 1:
    plan.tune(force_direction=int(sys.argv[1]))
bad = 0
for root in roots[:16]:
    plan.run(root); full, _ = plan.fetch(); full = full.copy(); L = plan.stats()["levels"]
    for k in range(max(1, L - 3), L + 1):
        for rep in range(4):
            plan.run(root, k)
            lv, _ = plan.fetch()
            st = plan.stats()
            ref = np.where((full >= 0) & (full <= k), full, -1)
            if not np.array_equal(lv, ref):
                bad += 1
                d = np.nonzero(lv != ref)[0]
                print("root", root, "k", k, "rep", rep, "ndiff", len(d), st)
                for v in d[:4]:
                    ins = tci[trp[v]:trp[v + 1]]
                    print("   v", v, "got", lv[v], "full", full[v], "in-neighbour levels(full)", sorted(set(full[ins].tolist()))[:6],
                          "in-neighbour levels(this run)", sorted(set(lv[ins].tolist()))[:6])
print("bad", bad)
