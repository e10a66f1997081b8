#!/usr/bin/env python3
"""A/B of the k-hop chain's engine options on the bench's own inputs (bench.py khop_inputs): ms per 1024-row batch, the
dominant kernels' in-library HIP-event times, and the (nnz, checksum, flops) triple that must not move.

  python tools/khop_sweep.py [--scale 24] [--batches 8] name=v1,v2,... [name=...]   # one-at-a-time sweeps
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from falkordb_amd import engine  # noqa: E402


def run(ctx, layers, batches, want_checksum=True):
    for b in batches[:2]:
        engine.expand_count(ctx, b, *layers, want_checksum=want_checksum)
    ctx.sync()
    t0 = time.perf_counter()
    tot = [0, 0, 0]
    for b in batches:
        n, c, f = engine.expand_count(ctx, b, *layers, want_checksum=want_checksum)
        tot[0] += n
        tot[1] = (tot[1] + (c or 0)) & 0xFFFFFFFFFFFFFFFF
        tot[2] += f
    dt = time.perf_counter() - t0
    ctx.prof_enable(True)
    for b in batches:
        engine.expand_count(ctx, b, *layers, want_checksum=want_checksum)
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    top = sorted(prof, key=lambda k: -k["ms"])[:4]
    return dt / len(batches) * 1e3, tuple(tot), [(k["kernel"], round(k["ms"] / max(k["launches"], 1), 3)) for k in top]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--dirty", action="store_true")
    ap.add_argument("sweeps", nargs="*")
    a = ap.parse_args()
    ctx = engine.Context(0)
    A, dp, dm, _ = bench.khop_inputs(ctx, a.scale, 16)
    srcs = bench.p_label_sources(A.nrows)
    batches = [srcs[i * 1024:(i + 1) * 1024] for i in range(a.batches)]
    layers = ([A] * 3, [dp] * 3, [dm] * 3) if a.dirty else ([A] * 3,)
    base = None
    out = []
    configs = [("default", None, None)]
    for sw in a.sweeps:
        name, vals = sw.split("=")
        configs += [(f"{name}={v}", name, int(v)) for v in vals.split(",")]
    for label, name, v in configs:
        if name:
            ctx.set_option(name, v)
        ms, tot, top = run(ctx, layers, batches)
        ms_c, _, _ = run(ctx, layers, batches, want_checksum=False)
        if base is None:
            base = tot
        rec = {"config": label, "ms_per_batch": round(ms, 3), "ms_per_batch_count_only": round(ms_c, 3),
               "same_result": tot == base, "top_kernels_ms_per_launch": top}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        if name:   # back to the default of this option before the next sweep
            pass
    print(json.dumps({"result": {"nnz": base[0], "checksum": f"{base[1]:016x}", "flops": base[2]}}))


if __name__ == "__main__":
    main()
