"""Whole-frontier fgpu_expand_count (spgemm.hip expand_count_scan) against the 1024-row calls it replaces: one call over the
first `nsrc` :P sources of the bench's R-MAT graph, for every (pass width, lanes, token) asked for — time, TEPS, and the sums
held equal to the per-batch calls' (nnz, flops) and to each other (checksum: the row hash is the row of the CALL, so it is
compared between the whole-frontier settings and, with --oracle, against oracle.expand_summary_omp over the same rows).
usage: python tools/scan_sweep.py [--scale 22] [--nsrc 32768] [--dirty] [--oracle] [--rows 1024,512] [--lanes 1,2,3,4] [--prof]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from falkordb_amd import engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--nsrc", type=int, default=32768)
    ap.add_argument("--hops", type=int, default=3)
    ap.add_argument("--dirty", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--rows", default="1024")
    ap.add_argument("--lanes", default="1,2,3,4")
    ap.add_argument("--prof", action="store_true", help="kernel table (HIP events in the library) of one call per setting")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from conftest import _BenchGraphs
    from test_gpu_scale import p_sources
    ctx = engine.Context(0)
    for o in args.opt:
        k, v = o.split("=")
        ctx.set_option(k, int(v))
    g = _BenchGraphs(ctx)
    A = ctx.mat_rmat(args.scale, 16, 0x5EED1234 + args.scale)
    n = A.nrows
    ids = np.arange(n, dtype=np.uint64)
    import oracle
    allp = ids[oracle.mix64(ids) % np.uint64(16) == 0]
    src = allp[:args.nsrc] if args.nsrc > 0 else allp
    layers = ([A] * args.hops,)
    host = None
    if args.dirty or args.oracle:
        g._g[args.scale] = (A, None, None)
    if args.dirty:
        rp, ci, _ = A.export_csr()
        g._g[args.scale] = (A, None, oracle.CSR(n, n, rp, ci))
        dp, dm, hdp, hdm = g.khop_layers(args.scale)
        layers = ([A] * args.hops, [dp] * args.hops, [dm] * args.hops)
        host = (g._g[args.scale][2], hdp, hdm)
    out = {"scale": args.scale, "nsrc": int(len(src)), "hops": args.hops, "dirty": args.dirty, "runs": []}
    # the calls it replaces: 1024 sources per call, one thread
    B = 1024
    nb = (len(src) + B - 1) // B
    engine.expand_count(ctx, src[:B], *layers)
    ctx.sync()
    t = time.perf_counter()
    bn = bf = 0
    for j in range(nb):
        a, _, f = engine.expand_count(ctx, src[j * B:(j + 1) * B], *layers)
        bn += a
        bf += f
    dt = time.perf_counter() - t
    out["per_batch_1024"] = {"calls": nb, "ms": round(dt * 1e3, 3), "TEPS": round(bf / dt, 1), "nnz": int(bn), "flops": int(bf)}
    print(json.dumps(out["per_batch_1024"]), flush=True)
    ref_cs = None
    if args.oracle:
        a = host[0] if host else None
        if a is None:
            rp, ci, _ = A.export_csr()
            a = oracle.CSR(n, n, rp, ci)
        lay = [(a, host[1], host[2]) if args.dirty else (a, None, None)] * args.hops
        t = time.perf_counter()
        r = oracle.expand_summary_omp(src, lay, chunk=64)
        out["oracle"] = {"nnz": int(r[0]), "checksum": int(r[1]), "flops": int(r[2]), "seconds": round(time.perf_counter() - t, 1)}
        print(json.dumps(out["oracle"]), flush=True)
        assert (r[0], r[2]) == (bn, bf), "per-batch sums differ from the oracle"
        ref_cs = int(r[1])
    first = None
    for rows in [int(x) for x in args.rows.split(",")]:
        for lanes in [int(x) for x in args.lanes.split(",")]:
            for token in (0,):
                ctx.set_option("expand_scan_rows", rows)
                ctx.set_option("expand_scan_lanes", lanes)
                got = engine.expand_count(ctx, src, *layers)             # warm: the lanes' pools
                best = None
                for _ in range(args.reps):
                    ctx.sync()
                    t = time.perf_counter()
                    got = engine.expand_count(ctx, src, *layers)
                    d = time.perf_counter() - t
                    best = d if best is None or d < best else best
                ok = (got[0], got[2]) == (bn, bf)
                if first is None:
                    first = got
                ok = ok and got == first and (ref_cs is None or got[1] == ref_cs)
                run = {"rows": rows, "lanes": lanes, "ms": round(best * 1e3, 3), "TEPS": round(got[2] / best, 1),
                       "passes": ctx.get_option("expand_scan_last_passes"), "live": ctx.get_option("expand_scan_last_live"),
                       "ms_per_1024_sources": round(best * 1e3 / (len(src) / 1024), 4), "ok": bool(ok)}
                if args.prof:
                    ctx.prof_enable(True)
                    engine.expand_count(ctx, src, *layers)
                    prof = ctx.prof_read()
                    ctx.prof_enable(False)
                    run["kernels"] = [{"kernel": k["kernel"], "ms": round(k["ms"], 3), "launches": k["launches"],
                                       "us_per_launch": round(k["ms"] / max(k["launches"], 1) * 1e3, 1)}
                                      for k in sorted(prof, key=lambda k: -k["ms"])[:14]]
                out["runs"].append(run)
                print(json.dumps(run), flush=True)
                assert ok, (got, first, bn, bf, ref_cs)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
