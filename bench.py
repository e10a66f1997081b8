#!/usr/bin/env python3
"""bench.py — TEPS of the boolean-vxm BFS on synthetic R-MAT graphs (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one whole BFS (the GrB_vxm frontier loop behind algo.BFS) from one of the 64
Graph500-style roots of a synthetic R-MAT graph resident in HBM.  N=1 runs BASELINE.json
configs[1] (RMAT scale-22, edge factor 16).  N>1 runs the same path on the column-slab
partition (one process per GPU, frontier all-gather over RCCL each level) with weak scaling:
scale = 22 + log2(N), i.e. the per-GPU edge count stays that of RMAT-22 (override: --scale).

Rank 0 prints ONE JSON line; `value` = total traversed edges (sum over BFS runs of the
out-degrees of reached vertices) / max-over-ranks wall time of the K steps.
Extra objects: `roofline` (dominant kernel, HIP-event timed in a second pass over the same
roots), `spmv_full_pass` (the north-star "RMAT-22 boolean SpMV" full-matrix pass) and
`cpu_baseline` (the CPU oracle's BFS timed on this box's host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(kernel, scale):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/traffic.json, written by tools/prof_bench.sh + tools/pmc_summary.py: FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM).  PMC needs rocprofv3 around the process, so it is not
    collected live; null when no profile of this workload is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(f"rmat{scale}", {}).get(kernel)
        return int(e["hbm_bytes_per_dispatch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def _cgroup_cpus():
    """CPUs the job may use per CFS period (cgroup v2 cpu.max or v1 cfs_quota/period); None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def pick_roots(A, want=64):
    """First `want` vertex ids with out-degree > 0."""
    roots, hi = [], 4096
    n = A.nrows
    while len(roots) < want:
        rows, _, _ = A.extract(0, min(hi, n) - 1)
        roots = np.unique(rows)[:want].tolist()
        if hi >= n:
            break
        hi *= 4
    return [int(r) for r in roots]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scale", type=int, default=0, help="R-MAT scale (default 22 + log2(gpus))")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--alpha", type=float, default=0.0, help="push->pull switch factor (0 = library default)")
    ap.add_argument("--force-dir", type=int, default=0, help="0 auto, 1 push only, 2 pull only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path even with one rank")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (fgpu_set_option)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch

    from falkordb_amd import dist as fdist
    from falkordb_amd import engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # --force-dist: drive the multi-rank code path (process group, column slab, slab backend, collectives)
    # with world_size 1 — a smoke test of the N > 1 path on a 1-GPU box, not a benchmark configuration
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as td
        td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    scale = args.scale or (22 + int(round(math.log2(world))))
    seed = 0x5EED1234 + scale
    ctx = engine.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    info = ctx.device_info()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    # ---- synthetic input, resident in HBM before anything is timed ----------------------
    t_build = time.time()
    A_full = ctx.mat_rmat(scale, args.edge_factor, seed)
    n, nnz = A_full.nrows, A_full.nvals
    roots = pick_roots(A_full, 64)
    if use_dist:
        lo, hi, slab = fdist.slab_range(n, rank, world)
        A = A_full.col_slab(lo, min(hi, n))
        A_full.free()
        At = A.transpose()
    else:
        A = A_full
        At = A.transpose()
    backend = fdist.HipSlabBackend(ctx, A, At, rank if use_dist else 0, world if use_dist else 1, dev)
    plan = backend.plan
    plan.tune(alpha=args.alpha, force_direction=args.force_dir)
    ctx.sync()
    t_build = time.time() - t_build

    def run_one(src):
        if use_dist:
            backend.run(src)
        else:
            plan.run(src, -1, False)

    # single GPU: two plans (workspaces) over the same matrices, so the host enqueues search i+1
    # while search i runs; every step is still one complete BFS and the stream executes them in order
    plans = [plan] if use_dist else [plan, engine.BfsPlan(ctx, A, At)]
    if not use_dist:
        plans[1].tune(alpha=args.alpha, force_direction=args.force_dir)

    def run_pipelined(srcs):
        for i, src in enumerate(srcs):
            plans[i % 2].run_async(src, -1, False, 0)   # levels=0: one more than this plan's previous search took
            if i > 0:
                plans[(i - 1) % 2].wait()
        if srcs:
            plans[(len(srcs) - 1) % 2].wait()

    # ---- untimed pass over every distinct root: per-root traversed-edge counts + warm-up --
    edges_by_root = {}
    stats_by_root = {}
    for r in roots:
        run_one(r)
        st = plan.stats()
        edges_by_root[r] = st["edges_traversed"]
        stats_by_root[r] = st
    if use_dist:
        t = torch.tensor([edges_by_root[r] for r in roots], dtype=torch.int64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.SUM)  # slab-local out-degree sums -> global
        for r, v in zip(roots, t.tolist()):
            edges_by_root[r] = int(v)
    if use_dist:
        for i in range(args.warmup):
            run_one(roots[i % len(roots)])
    else:
        run_pipelined([roots[i % len(roots)] for i in range(args.warmup)])

    # ---- timed region: exactly K steps ------------------------------------------------------
    def fence():
        if use_dist:
            td.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    if use_dist:
        for i in range(args.steps):
            run_one(roots[i % len(roots)])
    else:
        run_pipelined([roots[i % len(roots)] for i in range(args.steps)])
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())
    total_edges = sum(edges_by_root[roots[i % len(roots)]] for i in range(args.steps))
    teps = total_edges / dt

    # ---- roofline: per-kernel HIP-event timings over the same roots (second pass) ------------
    roofline = None
    spmv = None
    if not args.no_roofline and not use_dist:
        plan.profile(True)
        for i in range(args.steps):
            plan.run(roots[i % len(roots)], -1, False)
        prof = plan.profile_read()
        plan.profile(False)
        steps = [p for p in prof if p["launches"]]
        if steps:
            # The dominant kernel of the workload is bfs_fused_kernel: ONE kernel runs every level (the
            # <.., 1> / <.., 2> instantiations of the profiled pass only name a launch push / pull for the
            # profilers).  `roofline` is that kernel over all its level launches; the split by direction
            # (push levels are latency / atomic bound, pull levels stream column ids) is in `by_direction`.
            launches = sum(p["launches"] for p in steps)
            tot_ms = sum(p["ms"] for p in steps)
            tot_bytes = sum(p["alg_bytes"] for p in steps)
            per_launch_bytes = tot_bytes / launches
            per_launch_ms = tot_ms / launches
            ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            tr = [(pmc_traffic(p["kernel"].split(" (")[0], scale), p["launches"]) for p in steps]
            traffic = (int(sum(t * n for t, n in tr) / launches) if all(t is not None for t, _ in tr) else None)
            roofline = {"bound": "hbm", "kernel": "bfs_fused_kernel (every BFS level: push and pull launches)",
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": traffic,
                        "alg_bytes_per_launch": int(per_launch_bytes), "avg_launch_us": round(per_launch_ms * 1e3, 2),
                        "launches": int(launches),
                        "timing": "HIP events on the ctx stream around each launch of the profiled pass "
                                  "(same roots as the timed region)",
                        "by_direction": [{"kernel": p["kernel"], "ms_total": round(p["ms"], 4),
                                          "launches": int(p["launches"]),
                                          "avg_launch_us": round(p["ms"] / p["launches"] * 1e3, 2),
                                          "alg_bytes_per_launch": int(p["alg_bytes"] / p["launches"]),
                                          "traffic": pmc_traffic(p["kernel"].split(" (")[0], scale),
                                          "GBps": round(p["alg_bytes"] / max(p["ms"], 1e-9) / 1e6, 2),
                                          "frac": round(p["alg_bytes"] / max(p["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
                                         for p in steps],
                        "note": "BFS levels are latency / L2-line bound (bitmap probes, atomics); the HBM-bound "
                                "kernel of this path is the full-pass boolean SpMV reported in spmv_full_pass"}
        # the north-star full-matrix boolean SpMV pass (dense frontier, no mask, no early exit)
        # (LDS-tiled layout, tiled.hip) with the CSR pull kernel's figure beside it
        tinfo = At.build_tiles()
        ms, ab = engine.bench_spmv(ctx, At, which=2, iters=50)
        g = ab / (ms * 1e-3) / 1e9
        ms0, _ = engine.bench_spmv(ctx, At, which=0, iters=10)
        spmv = {"kernel": "tiled_mxv_kernel", "avg_launch_us": round(ms * 1e3, 2), "alg_bytes": int(ab),
                "achieved": round(g, 2), "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": round(g / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic("tiled_mxv_kernel", scale),
                "layout": {k: tinfo[k] for k in ("tile_bits", "tiles", "items", "entries", "vec", "k", "bytes")},
                "csr_pull_us": round(ms0 * 1e3, 2), "csr_pull_GBps": round(ab / (ms0 * 1e-3) / 1e9, 2)}

    # ---- CPU baseline on this box's host cores, bounded sample, rank 0 / N=1 only -------------------
    # The reference's path is LAGraph's push/pull BFS over SuiteSparse:GraphBLAS with OpenMP inside every
    # vxm / mxv; neither library exists in this image, so the stand-in is the oracle's OpenMP
    # direction-optimizing BFS (oracle/oracle_omp.c, same algorithm family) on every host core, with the
    # serial queue BFS (oracle/oracle.c) beside it.  Baseline only: the roofline fraction is the quality bar.
    cpu = None
    if not args.no_cpu_baseline and not use_dist and rank == 0:
        import oracle
        rp, ci, _ = A.export_csr()
        a = oracle.CSR(n, n, rp, ci)
        trp, tci, _ = At.export_csr()
        at = oracle.CSR(n, n, trp, tci)
        # thread count: the box reports every hardware thread of the host, but a job usually owns fewer
        # (cgroup quota) and an oversubscribed OpenMP team is slower than one thread — calibrate on one root
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = _cgroup_cpus()
        cap = min(ncpu, max(1, int(quota * 2))) if quota else ncpu     # a team beyond the CFS quota only gets throttled
        oracle.bfs_omp(a, at, roots[0], -1, threads=1)                  # warm-up: page faults
        best = (0.0, 1)
        calib = {}
        for th in sorted({t for t in (1, 2, 4, 8, 16, 32, 64, 128, cap) if t <= cap}):
            rates = []
            for rep in range(3):                                        # median of three: single probes are noisy on a shared host
                t1 = time.perf_counter()
                _, e = oracle.bfs_omp(a, at, roots[rep % len(roots)], -1, threads=th)
                rates.append(e / (time.perf_counter() - t1))
            rate = sorted(rates)[1]
            calib[th] = round(rate / 1e6, 1)
            if rate > best[0]:
                best = (rate, th)
            elif rate < 0.25 * best[0]:
                break                                                  # far past the knee: stop trying larger teams
        threads = best[1]
        oracle.bfs_omp(a, at, roots[0], -1, threads=threads)
        e_cpu, t_cpu, k = 0, 0.0, 0
        budget = args.cpu_seconds * 0.7
        while t_cpu <= budget:                                          # cycle the 64 roots until the budget is spent
            r = roots[k % len(roots)]
            t1 = time.perf_counter()
            _, e = oracle.bfs_omp(a, at, r, -1, threads=threads)
            t_cpu += time.perf_counter() - t1
            e_cpu += e
            k += 1
        e_ser, t_ser, ks = 0, 0.0, 0
        for r in roots:
            t1 = time.perf_counter()
            _, _, e = oracle.bfs(a, r, -1, want_parent=False)
            t_ser += time.perf_counter() - t1
            e_ser += e
            ks += 1
            if t_ser > args.cpu_seconds * 0.3:
                break
        cpu = {"value": round(e_cpu / t_cpu, 1), "unit": "TEPS", "cores": threads, "kind": "port",
               "sample": f"{k} BFS runs cycling the 64 roots of the same RMAT-{scale} graph, {t_cpu:.1f} s, OpenMP push/pull BFS "
                         f"(oracle/oracle_omp.c orc_bfs_omp) on {threads} threads; CPU stand-in for "
                         f"LAGraph + SuiteSparse:GraphBLAS, which are absent from this image",
               "threads_calibration_MTEPS": calib, "host_cpus_visible": ncpu, "cgroup_cpu_quota": quota,
               "serial": {"value": round(e_ser / t_ser, 1), "cores": 1,
                          "sample": f"{ks} roots, {t_ser:.1f} s, serial queue BFS (oracle/oracle.c orc_bfs)"}}

    if rank == 0:
        st0 = stats_by_root[roots[0]]
        out = {
            "metric": "traversed edges/sec (TEPS) on BFS (boolean vxm frontier loop), synthetic R-MAT",
            "value": round(teps, 1),
            "unit": "TEPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"RMAT scale-{scale} BFS (boolean GrB_vxm frontier loop), edge factor {args.edge_factor}, "
                            f"64 Graph500-style roots, directed, deduplicated",
                "scale": scale, "vertices": int(n), "edges": int(nnz),
                "parallelism": ("1 GPU" if world == 1 else f"column-slab x{world} + frontier all-gather (RCCL)"),
                "direction": {0: "auto push/pull", 1: "push only", 2: "pull only"}[args.force_dir],
                "device": info["name"], "build_seconds": round(t_build, 2),
                "root0_levels": st0["levels"], "root0_push_levels": st0["push_levels"],
                "root0_pull_levels": st0["pull_levels"],
            },
            "roofline": roofline,
            "spmv_full_pass": spmv,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
