"""k-hop batches from T host threads sharing one context (a lane = stream + pool + staging per thread): the host-side
synchronisations of one batch's chain overlap the kernels of another's.  usage: python tools/experiments/khop_threads.py [scale] [batches]"""
import sys, time, json
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, ".")
import bench
from falkordb_amd import engine

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = engine.Context(0)
A, dp, dm, _ = bench.khop_inputs(ctx, scale, 16)
srcs = bench.p_label_sources(A.nrows)
batches = [srcs[i * 1024:(i + 1) * 1024] for i in range(nb)]
for name, layers in (("clean", ([A] * 3,)), ("dirty", ([A] * 3, [dp] * 3, [dm] * 3))):
    ref = None
    for T in (1, 2, 3):
        def one(b):
            return engine.expand_count(ctx, b, *layers, want_checksum=True)
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(one, batches[:2 * T]))          # warm every lane (pool, staging, first-use indexes)
            ctx.sync()
            t0 = time.perf_counter()
            res = list(ex.map(one, batches))
            dt = time.perf_counter() - t0
        tot = (sum(r[0] for r in res), sum(r[1] for r in res) & 0xFFFFFFFFFFFFFFFF, sum(r[2] for r in res))
        if ref is None: ref = tot
        print(json.dumps({"scale": scale, "layers": name, "threads": T, "ms_per_batch": round(dt / nb * 1e3, 3),
                          "TEPS": round(tot[2] / dt / 1e12, 3), "same_result": tot == ref}), flush=True)
