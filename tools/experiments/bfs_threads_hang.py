#!/usr/bin/env python3
"""Where do >= 3 query threads driving BFS plans stall?  Python stacks of every thread after 12 s (faulthandler)."""
import faulthandler, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from falkordb_amd import engine
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
faulthandler.dump_traceback_later(12, exit=True)
ctx = engine.Context(0)
A = ctx.mat_rmat(22, 16, 0x5EED1234 + 22)
At = A.transpose()
roots = bench.pick_roots(A, 64)
ctx.sync()
prog = [0] * k
allp = {}
def work(t):
    ps = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
    allp[t] = ps
    print("thread", t, "plans made", flush=True)
    for i in range(64):
        ps[i % 2].run_async(roots[(t + i * k) % 64], -1, False, 0)
        if i > 0:
            ps[(i - 1) % 2].wait()
        prog[t] = i
    ps[1].wait()
    ctx.sync()
    print("thread", t, "done", flush=True)
ths = [threading.Thread(target=work, args=(t,)) for t in range(k)]
for th in ths: th.start()
t0 = time.time()
while any(th.is_alive() for th in ths) and time.time() - t0 < 9:
    time.sleep(1); print("progress", prog, flush=True)
    if time.time() - t0 > 5:
        for t, th in enumerate(ths):
            if th.is_alive():
                for j, pl in enumerate(allp.get(t, [])):
                    st = pl.stats()
                    print("  stuck thread", t, "plan", j, {k_: st[k_] for k_ in ("levels", "reached", "push_levels", "pull_levels", "last_frontier")}, flush=True)
for th in ths: th.join()
print("all done")
faulthandler.cancel_dump_traceback_later()
