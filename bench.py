#!/usr/bin/env python3
"""bench.py — TEPS of the boolean-vxm BFS and of the k-hop MATCH chain on synthetic R-MAT graphs (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one whole BFS (the GrB_vxm frontier loop behind algo.BFS) from one of the 64
Graph500-style roots of a synthetic R-MAT graph resident in HBM.  N=1 runs BASELINE.json
configs[1] (RMAT scale-22, edge factor 16).  N>1 runs the same path on a column-slab partition
balanced by nnz (one process per GPU, one frontier all-gather-v over RCCL/xGMI per level).

N > 1 runs RMAT-26 (BASELINE config 4: the same graph at every N, strong scaling, `--gpus 1 --scale 26` is the base
point); the level loop and the per-level frontier exchange run inside libfgpu.so (fgpu_bfs_dist_run over an RCCL
communicator the library owns), torch.distributed only launches the ranks, carries the communicator id and fences.

Rank 0 prints ONE JSON line; `value` = total traversed edges (sum over BFS runs of the
out-degrees of reached vertices) / max-over-ranks wall time of the K steps.
Extra objects: `roofline` (dominant kernel, HIP-event timed in a second pass over the same
roots), `spmv_full_pass` (the north-star "RMAT-22 boolean SpMV" full-matrix pass), `khop_match`
(BASELINE config 3: RMAT-24 3-hop MATCH as a masked GrB_mxm chain — CondTraverseOp::expand_batch's
device core, cond_traverse.rs:452-751 / matrix.rs:1317-1402 — clean and dirty layers, with its own
roofline object and CPU baseline) and `cpu_baseline` (the CPU oracle's BFS timed on this box's host
cores, bounded sample).  `traffic` figures are HBM bytes per launch from rocprofv3 --pmc passes that
bench.py runs over a reduced replay of the same workload (`--pmc-child`) at the end of the run.
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


def committed_traffic(kernel, scale):
    """Fallback when the live PMC passes cannot run: HBM bytes per launch from the committed rocprofv3 --pmc passes
    of this same command (profiles/traffic.json, tools/prof_bench.sh) — only if that file was taken from the
    kernel sources as they are now (it records their hash)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("_csrc_sha256") != csrc_hash():
            return None
        e = t.get(f"rmat{scale}", {}).get(kernel)
        return int(e["hbm_bytes_per_dispatch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def csrc_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "falkordb_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


PMC_GROUPS = (  # (reported name, regex over rocprofv3's Kernel_Name)
    ("bfs_fused_kernel (push level)", r"bfs_fused_kernel<(true|false), 1>"),
    ("bfs_fused_kernel (pull level)", r"bfs_fused_kernel<(true|false), 2>"),
    ("bfs_fused_kernel (blind loop)", r"bfs_fused_kernel<(true|false), 0>"),
    ("tiled_mxv_kernel", r"tiled_mxv_kernel"),
    ("blocked_mxv_kernel", r"blocked_mxv_kernel"),
    ("bp_pull_kernel<dense>", r"bp_pull_kernel<\d+, false, 0>"),
    ("bp_pull_kernel<sparse>", r"bp_pull_kernel<\d+, true, 0>"),
    ("bp_pull_kernel<dense, count>", r"bp_pull_kernel<\d+, false, [12]>"),
    ("bp_pull_kernel<sparse, count>", r"bp_pull_kernel<\d+, true, [12]>"),
    ("bp_count_kernel<checksum>", r"bp_count_kernel<true>"),
    ("bp_count_kernel<count>", r"bp_count_kernel<false>"),
    ("bp_delta_kernel<dm>", r"bp_delta_kernel<true>"),
    ("bp_delta_kernel<dp>", r"bp_delta_kernel<false>"),
    ("bp_pull_groups_kernel", r"bp_pull_groups_kernel"),
    ("bp_rows_kernel<emit>", r"bp_rows_kernel<true>"),
    ("bp_rows_kernel<count>", r"bp_rows_kernel<false>"),
)


def live_pmc(args, timeout_s=420):
    """HBM bytes per launch of the hot kernels, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE cannot share a pass: TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots") over
    `bench.py --pmc-child`, a reduced replay of this run's workloads (same graphs, same kernels, a few steps).
    FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request of a wide stream, so the read
    side is doubled (MI355X_MICROARCH.md §HBM).  Returns {name: {...}} or {"error": ...}."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = tempfile.mkdtemp(prefix="fgpu_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--scale", str(args.scale or 22),
             "--khop-scale", str(args.khop_scale)]
    raw = {}
    t0 = time.time()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True,
                               timeout=max(30, timeout_s - (time.time() - t0)))
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {ctr} exited {r.returncode}: {r.stderr[-300:]}"}
            acc = collections.defaultdict(lambda: [0.0, 0])
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    for name, rx in PMC_GROUPS:
                        if re.search(rx, row["Kernel_Name"]):
                            acc[name][0] += float(row["Counter_Value"])
                            acc[name][1] += 1
                            break
            for name, (tot, n) in acc.items():
                raw.setdefault(name, {})[ctr] = (tot / n * 1024.0, n)
    except subprocess.TimeoutExpired:
        return {"error": f"rocprofv3 passes exceeded {timeout_s} s"}
    except OSError as e:
        return {"error": str(e)}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {}
    for name, d in raw.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            f, nf = d["FETCH_SIZE"]
            w, _ = d["WRITE_SIZE"]
            res[name] = {"fetch_bytes_raw": int(f), "fetch_bytes_x2": int(2 * f), "write_bytes": int(w),
                         "hbm_bytes_per_dispatch": int(2 * f + w), "dispatches": nf}
    res["_seconds"] = round(time.time() - t0, 1)
    return res


def probe_reference_libs():
    """look for the reference's own CPU libraries on this box before falling back to the port.  They
    are not part of this image (no SuiteSparse:GraphBLAS, no LAGraph, no python-graphblas), so the expected answer is
    "absent" — recorded in the bench line so the label "port" is checked, not assumed."""
    import ctypes.util
    import glob
    found = {}
    dirs = ("/usr/lib", "/usr/lib64", "/usr/local/lib", "/usr/lib/x86_64-linux-gnu", "/opt/lib", "/opt/local/lib")
    for lib in ("graphblas", "lagraph", "lagraphx"):
        path = ctypes.util.find_library(lib)          # ldconfig cache + the linker's search path
        if not path:
            hits = [h for d in dirs for h in glob.glob(os.path.join(d, f"lib{lib}.so*"))][:1]
            path = hits[0] if hits else None
        found["lib" + lib] = path
    try:
        import importlib.util
        found["python_graphblas"] = importlib.util.find_spec("graphblas") is not None
    except Exception:
        found["python_graphblas"] = False
    hdr = [h for h in ("/usr/include/GraphBLAS.h", "/usr/include/suitesparse/GraphBLAS.h", "/usr/local/include/GraphBLAS.h",
                       "/usr/local/include/suitesparse/GraphBLAS.h") if os.path.exists(h)]
    found["GraphBLAS.h"] = hdr[0] if hdr else None
    return found


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


def mix64_np(z):
    z = np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def p_label_sources(n):
    """The synthetic label :P of SURVEY.md §8d: ids with hash(id) % 16 == 0 (hash = splitmix64 finaliser), ascending."""
    ids = np.arange(n, dtype=np.uint64)
    return ids[mix64_np(ids) % np.uint64(16) == 0]


def khop_inputs(ctx, scale, edge_factor, want_host=False):
    """RMAT-<scale> adjacency + one dirty layer pair: dm = a uniformly random 0.1 % of the stored entries
    (fgpu_mat_sample), dp = as many uniformly random coordinates outside the matrix (the Delta invariants
    dm ⊆ m, dp ∩ m = ∅, versioned_matrix.rs:214-235).  want_host: also the three layers as oracle.CSR (the parity
    check's inputs, exported before the delta layers are made hypersparse)."""
    A = ctx.mat_rmat(scale, edge_factor, 0x5EED1234 + scale)
    n = A.nrows
    dm = A.sample(0xD3170 + scale, 1000)
    rng = np.random.default_rng(0xADD5 + scale)
    k = max(1, A.nvals // 1000)
    raw = ctx.mat_from_coo(n, n, rng.integers(0, n, k, dtype=np.uint64), rng.integers(0, n, k, dtype=np.uint64))
    dp = raw.merge(None, A)
    raw.free()
    host = None
    if want_host:
        import oracle
        host = tuple(oracle.CSR(m.nrows, m.ncols, *m.export_csr()[:2]) for m in (A, dp, dm))

    def hypersparse(m):
        # Delta layers are hypersparse in the reference (Delta<T>::new pins them so, versioned_matrix.rs:214-235): a row list
        # and a short row-pointer array instead of N + 1 row pointers for ~10^5 stored rows
        rp, ci, _ = m.export_csr()
        deg = np.diff(rp.astype(np.int64))
        rows = np.nonzero(deg)[0].astype(np.uint64)
        short = np.concatenate([[0], np.cumsum(deg[deg > 0])]).astype(np.uint64)
        h = ctx.mat_from_csr(n, n, short, ci, hyper_rows=rows)
        m.free()
        return h
    if os.environ.get("FGPU_BENCH_HYPER_DELTAS", "1") != "0":
        dp, dm = hypersparse(dp), hypersparse(dm)
    return A, dp, dm, host


def khop_alg_bytes(rows, hop_nnz, flops, mask_nnz=0):
    """SURVEY.md §8d, one ANY_PAIR SpGEMM hop C<¬M> = F·A: 4(r+1) + 4 nnz(F) (read F) + 8 nnz(F) (A row-pointer
    pairs) + 4 flops (A column ids gathered) + 4 nnz(M) + 4(r+1) + 4 nnz(C) (write C); the chain is the sum."""
    f_nnz = [rows] + list(hop_nnz[:-1])
    return sum(8 * (rows + 1) + 12 * f + 4 * c for f, c in zip(f_nnz, hop_nnz)) + 4 * flops + 4 * mask_nnz


def cpu_threads():
    """Threads for the CPU legs: the job's CPU quota (cgroup), not the host's hardware-thread count — an OpenMP team
    beyond the quota is throttled by CFS and its rate becomes noise (4 thr 1433, 8 thr 992, 16 thr 2073,
    32 thr 1306 MTEPS on a 16-CPU quota)."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpus()
    return max(1, min(ncpu, int(quota))) if quota else ncpu, ncpu, quota


def khop_leg(ctx, engine, args, scale, nb_want, graph=None, parity_rows=1024):
    """BASELINE config 3's shape at one scale: RMAT-<scale> 3-hop MATCH (a:P)-->()-->()-->(c) as a masked GrB_mxm chain =
    the device core of CondTraverseOp::expand_batch (cond_traverse.rs:452-751: F = build(sources); F = delta_lmxm(F; hop)
    per hop, matrix.rs:1317-1402), batches of 1024 :P sources, result = count + order-independent checksum on the device
    (the (row, dest) stream of 1.5 G entries per batch at scale 24 does not fit a host buffer; the materialised form is
    khop_emit_leg).  t = wall time of the fgpu_expand_count calls, H2D of the sources and D2H of the results included;
    matrices resident.  Parity: batch 0's (nnz, checksum, flops), clean AND dirty, against the oracle's delta_lmxm
    chain (oracle.expand_summary_omp, `parity_rows` of the batch's rows) — a mismatch aborts the bench line."""
    hops, B = 3, 1024
    t0 = time.time()
    A, dp, dm, host = graph if graph is not None else khop_inputs(ctx, scale, args.edge_factor, want_host=not args.no_parity)
    ctx.sync()
    t_build = time.time() - t0
    n, nnz = A.nrows, A.nvals
    srcs = p_label_sources(n)
    nb_all = len(srcs) // B
    nb = nb_all if nb_want <= 0 else min(nb_want, nb_all)
    batches = [srcs[i * B:(i + 1) * B] for i in range(nb)]
    out = {"workload": f"RMAT scale-{scale} {hops}-hop MATCH (a:P)-->()-->()-->(c): CondTraverse expand_batch core "
                       f"(masked GrB_mxm ANY_PAIR chain), sources = label :P (hash(id) % 16 == 0), batches of {B}",
           "scale": scale, "vertices": int(n), "edges": int(nnz), "hops": hops, "batch_rows": B,
           "label_P_sources": int(len(srcs)), "batches_timed": nb,
           "sample": (f"the first {nb} of {nb_all} batches of the :P set in ascending id order" if nb < nb_all
                      else "the whole :P set"),
           "result": "count + checksum on device (fgpu_expand_count)", "build_seconds": round(t_build, 2),
           "nnz_dp": int(dp.nvals), "nnz_dm": int(dm.nvals)}
    prof_tables = {}
    batch0 = {}
    for name, layers in (("clean", ([A] * hops, None, None)), ("dirty", ([A] * hops, [dp] * hops, [dm] * hops))):
        for b in batches[:2]:                                   # warm-up: transpose cache, item lists, pools
            engine.expand_count(ctx, b, *layers)
        ctx.sync()
        t1 = time.perf_counter()
        tot_f = tot_n = 0
        cs = 0
        for i, b in enumerate(batches):
            nn, c, f = engine.expand_count(ctx, b, *layers)
            if i == 0:
                batch0[name] = (nn, c, f)
            tot_n += nn
            tot_f += f
            cs = (cs + c) & 0xFFFFFFFFFFFFFFFF
        dt = time.perf_counter() - t1
        t1 = time.perf_counter()
        for b in batches:
            engine.expand_count(ctx, b, *layers, want_checksum=False)
        dt_count = time.perf_counter() - t1
        # kernel table: the same calls again with HIP events around every modelled launch
        ctx.prof_enable(True)
        t1 = time.perf_counter()
        for b in batches:
            engine.expand_count(ctx, b, *layers)
        dt_prof = time.perf_counter() - t1
        prof = ctx.prof_read()
        ctx.prof_enable(False)
        prof_tables[name] = prof
        # per-hop result sizes (untimed) for the §8d byte count of the SpGEMM form
        alg = 0
        hop_tot = [0] * hops
        nlv = min(nb, 4)
        for b in batches[:nlv]:
            lv = engine.expand_levels(ctx, b, *layers)
            hn = [int(x) for x in lv["hop_nnz"]]
            alg += khop_alg_bytes(B, hn, int(lv["flops"]))
            hop_tot = [a_ + b_ for a_, b_ in zip(hop_tot, hn)]
        out[name] = {"ms_per_batch": round(dt / nb * 1e3, 3), "TEPS": round(tot_f / dt, 1),
                     "flops": int(tot_f), "out_nnz": int(tot_n), "checksum": f"{cs:016x}",
                     "hop_nnz_per_batch": [h // nlv for h in hop_tot],
                     "spgemm_form_alg_bytes_per_batch": int(alg // nlv),
                     "count_only": {"ms_per_batch": round(dt_count / nb * 1e3, 3), "TEPS": round(tot_f / dt_count, 1)},
                     "ms_per_batch_with_kernel_events": round(dt_prof / nb * 1e3, 3)}
    out["note"] = ("spgemm_form_alg_bytes follows SURVEY.md §8d's SpGEMM row (4 B per traversed edge + F / C / row-pointer "
                   "terms): what a gather-and-sort product of the same chain would move.  Dense hops run in bit form (one "
                   "pass over A' per hop whatever the traversed-edge count), so no fraction of peak is quoted against that "
                   "figure — the kernel-level roofline (compulsory bytes of the bit form, live traffic) is `roofline` below")
    kern = sorted(prof_tables["clean"], key=lambda k: -k["ms"])
    out["kernels"] = {name: [{"kernel": k["kernel"], "ms_total": round(k["ms"], 3), "launches": k["launches"],
                              "avg_launch_us": round(k["ms"] / max(k["launches"], 1) * 1e3, 2),
                              "alg_bytes_per_launch": int(k["alg_bytes"] / max(k["launches"], 1)),
                              "GBps": round(k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6, 1)}
                             for k in sorted(tab, key=lambda k: -k["ms"])[:10]] for name, tab in prof_tables.items()}
    if kern:
        d = kern[0]
        ach = d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6
        out["roofline"] = {"bound": "hbm", "kernel": d["kernel"], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                           "alg_bytes_per_launch": int(d["alg_bytes"] / d["launches"]),
                           "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2), "launches": d["launches"],
                           "share_of_kernel_time": round(d["ms"] / max(sum(k["ms"] for k in kern), 1e-9), 3),
                           "timing": "HIP events around each launch (fgpu_prof_*), clean layers, the timed batches replayed"}
    # ---- parity of what was just timed, and the CPU baseline from the same oracle run -------------------------
    if not args.no_parity and host is not None:
        import oracle
        a, hdp, hdm = host
        threads, ncpu, quota = cpu_threads()
        rows = batches[0][:parity_rows]
        gpu = dict(batch0)
        if parity_rows < B:                                   # (scale 26: the oracle takes a slice of batch 0)
            gpu = {"clean": engine.expand_count(ctx, rows, [A] * hops),
                   "dirty": engine.expand_count(ctx, rows, [A] * hops, [dp] * hops, [dm] * hops)}
        t1 = time.perf_counter()
        ref_clean = oracle.expand_summary_omp(rows, [(a, None, None)] * hops, chunk=64, threads=threads)
        t_clean = time.perf_counter() - t1
        t1 = time.perf_counter()
        ref_dirty = oracle.expand_summary_omp(rows, [(a, hdp, hdm)] * hops, chunk=64, threads=threads)
        t_dirty = time.perf_counter() - t1
        ok = tuple(gpu["clean"]) == tuple(ref_clean[:3]) and tuple(gpu["dirty"]) == tuple(ref_dirty[:3])
        out["parity"] = {"checked": True, "ok": bool(ok), "rows": int(len(rows)),
                         "what": f"(nnz, checksum, flops) of batch 0{'' if parity_rows >= B else ' rows [0, %d)' % parity_rows}, "
                                 "clean and dirty layers, fgpu_expand_count vs the oracle's delta_lmxm chain "
                                 "(oracle.expand_summary_omp, matrix.rs:1317-1402)",
                         "clean": {"nnz": int(ref_clean[0]), "checksum": f"{ref_clean[1]:016x}", "flops": int(ref_clean[2])},
                         "dirty": {"nnz": int(ref_dirty[0]), "checksum": f"{ref_dirty[1]:016x}", "flops": int(ref_dirty[2])}}
        if not ok:
            raise SystemExit(f"bench.py: k-hop parity FAILED at scale {scale}: gpu {gpu} vs oracle clean {ref_clean[:3]} "
                             f"dirty {ref_dirty[:3]}")
        out["cpu_baseline"] = {"value": round(ref_clean[2] / t_clean, 1), "unit": "TEPS", "cores": threads, "kind": "port",
                               "sample": f"{len(rows)} of batch 0's 1024 :P sources, 3 hops, clean layers, {t_clean:.1f} s "
                                         f"(dirty layers: {t_dirty:.1f} s = {ref_dirty[2] / t_dirty / 1e9:.2f} GTEPS), row-parallel "
                                         f"Gustavson ANY_PAIR products (oracle/oracle_omp.c orc_mxm_omp) on {threads} threads "
                                         f"(cgroup quota {quota}, {ncpu} hardware threads visible); CPU stand-in for "
                                         f"SuiteSparse:GraphBLAS GrB_mxm, which is absent from this image",
                               "flops": int(ref_clean[2]), "out_nnz": int(ref_clean[0])}
    else:
        out["parity"] = {"checked": False}
    return out, (A, dp, dm, host, batches)


def khop_emit_leg(ctx, engine, args, A, host, batches):
    """The MATERIALISED form of the same chain: fgpu_expand_mat returns F on the device as the
    (rowptr, dest) CSR the operator then walks (cond_traverse.rs:608, 644-751) — (i) the 2-hop chain of a 1024-row batch
    (32 M entries per batch at RMAT-24), (ii) the 3-hop chain of 64 rows (90 M entries).  Timed: the call, result on the
    device; beside it the host-array entry fgpu_expand (PCIe + widening to 64-bit ids included).  Batch 0 of each is
    compared entry for entry with the oracle's chain."""
    import oracle
    out = {}
    threads = cpu_threads()[0]
    for name, hops, rows, nb in (("two_hop_1024_rows", 2, 1024, min(8, len(batches))),
                                 ("three_hop_64_rows", 3, 64, min(8, len(batches)))):
        bl = [b[:rows] for b in batches[:nb]]
        for b in bl[:2]:
            m_, _ = engine.expand_mat(ctx, b, [A] * hops)
            m_.free()
        ctx.sync()
        t1 = time.perf_counter()
        tot_n = tot_f = 0
        for b in bl:
            m_, f = engine.expand_mat(ctx, b, [A] * hops)
            tot_n += m_.nvals
            tot_f += f
            m_.free()
        ctx.sync()
        dt = time.perf_counter() - t1
        ctx.prof_enable(True)
        for b in bl:
            m_, f = engine.expand_mat(ctx, b, [A] * hops)
            m_.free()
        prof = ctx.prof_read()
        ctx.prof_enable(False)
        t1 = time.perf_counter()
        rp, dest, _ = engine.expand(ctx, bl[0], [A] * hops)
        dt_host = time.perf_counter() - t1
        rec = {"hops": hops, "rows": rows, "batches": nb, "ms_per_batch": round(dt / nb * 1e3, 3),
               "TEPS": round(tot_f / dt, 1), "out_nnz_per_batch": int(tot_n // nb),
               "host_arrays_ms_first_batch": round(dt_host * 1e3, 3),
               "kernels": [{"kernel": k["kernel"], "ms_total": round(k["ms"], 3), "launches": k["launches"],
                            "avg_launch_us": round(k["ms"] / max(k["launches"], 1) * 1e3, 2),
                            "alg_bytes_per_launch": int(k["alg_bytes"] / max(k["launches"], 1)),
                            "GBps": round(k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6, 1),
                            "frac": round(k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
                           for k in sorted(prof, key=lambda k: -k["ms"])[:8]]}
        if not args.no_parity and host is not None:
            c, fl, _ = oracle.expand_omp(bl[0], [(host[0], None, None)] * hops, threads=threads)
            ok = bool(np.array_equal(rp, c.rowptr) and np.array_equal(dest, c.colidx))
            rec["parity"] = {"checked": True, "ok": ok, "what": "batch 0's (rowptr, dest) arrays vs the oracle's chain, entry for entry",
                             "nnz": int(c.nnz)}
            if not ok:
                raise SystemExit(f"bench.py: materialised k-hop parity FAILED ({name})")
            del c
        out[name] = rec
    return out


def varlen_leg(ctx, engine, args, rank=0, world=1):
    """BASELINE config 5's stand-in (~0.5 M vertices / ~20 M edges = R-MAT scale 19, edge factor 38),
    `[*1..4]` reachability from 1024-source batches with 0.1 % tombstones + pending adds: fgpu_expand_levels (per-hop
    frontier sets + the DISTINCT union).  Sources are sharded over the ranks (rank r takes batches r, r + world, ...), A
    replicated: no collective on this path.  Returns this rank's (flops, seconds, batches)."""
    A, dp, dm, _ = khop_inputs(ctx, 19, 38, want_host=False)
    srcs = p_label_sources(A.nrows)
    B = 1024
    nb_all = len(srcs) // B
    mine = [srcs[i * B:(i + 1) * B] for i in range(rank, nb_all, world)][:8]
    layers = ([A] * 4, [dp] * 4, [dm] * 4)
    for b in mine[:1]:
        engine.expand_levels(ctx, b, *layers)
    ctx.sync()
    t1 = time.perf_counter()
    fl = un = 0
    for b in mine:
        lv = engine.expand_levels(ctx, b, *layers)
        fl += int(lv["flops"])
        un += int(lv["union_nnz"])
    dt = time.perf_counter() - t1
    dp.free(); dm.free(); A.free()
    return fl, dt, len(mine), un


def spmv_pass(ctx, engine, At, scale, iters=50):
    """The north-star roofline case: one full-matrix boolean pull pass (dense frontier, no mask, no early exit) over the
    LDS-tile layout of A', HIP-event timed per launch.  `warm`: back-to-back passes (at RMAT-22 the 268 MB layout sits in
    the 256 MiB Infinity Cache between passes — MI355X_MICROARCH.md: "scale past L3 before reading FETCH_SIZE");
    `cold`: 512 MiB of scratch is READ before every timed pass (no dirty lines left behind), so the layout streams from HBM.  `frac` (the
    headline of this object) is the COLD figure; the CSR pull of the same pass is beside it."""
    tinfo = At.build_tiles()
    ms, ab = engine.bench_spmv(ctx, At, which=2, iters=iters)
    ms_c, _ = engine.bench_spmv(ctx, At, which=3, iters=max(8, iters // 4))
    ms0, _ = engine.bench_spmv(ctx, At, which=0, iters=10)

    def gb(ms_):
        return ab / (ms_ * 1e-3) / 1e9
    kern = "blocked_mxv_kernel" if tinfo["tile_bits"] == 18 else "tiled_mxv_kernel"    # (blocked.hip's tiles are 2^18 columns)
    return {"kernel": kern, "scale": scale, "alg_bytes": int(ab),
            "avg_launch_us": round(ms_c * 1e3, 2), "achieved": round(gb(ms_c), 2), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(gb(ms_c) / HBM_PEAK_GBS, 4), "cache_state": "cold (512 MiB of scratch read before every timed pass)",
            "warm": {"avg_launch_us": round(ms * 1e3, 2), "achieved": round(gb(ms), 2), "frac": round(gb(ms) / HBM_PEAK_GBS, 4),
                     "cache_state": "back-to-back passes (Infinity Cache holds what fits of the layout)"},
            "traffic": None,
            "layout": {k: tinfo[k] for k in ("tile_bits", "tiles", "items", "entries", "vec", "k", "bytes")},
            "csr_pull_us": round(ms0 * 1e3, 2), "csr_pull_GBps": round(gb(ms0), 2)}


def strong_scaling_base(ctx, engine, args, scale=26, steps=32, warmup=8):
    """The N = 1 point of the curve `--gpus N` (N > 1) measures: the same RMAT-26 BFS on this one device, same 64-root
    rule, same two-plan pipelined loop as the headline.  Reported inside the N = 1 line so that the driver's N = 1, 2, 4,
    8 series has its base point on the same graph."""
    t0 = time.time()
    A = ctx.mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
    At = A.transpose()
    roots = pick_roots(A, 64)
    plans = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
    for p in plans:
        p.tune(alpha=args.alpha, force_direction=args.force_dir)
    ctx.sync()
    t_build = time.time() - t0
    edges = {}
    for r in roots[:max(steps, warmup)]:
        plans[0].run(r, -1, False)
        edges[r] = plans[0].stats()["edges_traversed"]

    def pipelined(srcs):
        for i, src in enumerate(srcs):
            plans[i % 2].run_async(src, -1, False, 0)
            if i > 0:
                plans[(i - 1) % 2].wait()
        plans[(len(srcs) - 1) % 2].wait()

    pipelined([roots[i % len(roots)] for i in range(warmup)])
    ctx.sync()
    t1 = time.perf_counter()
    pipelined([roots[i % len(roots)] for i in range(steps)])
    ctx.sync()
    dt = time.perf_counter() - t1
    tot = sum(edges[roots[i % len(roots)]] for i in range(steps))
    out = {"workload": f"RMAT scale-{scale} BFS on 1 GPU (base point of the `--gpus N` strong-scaling curve, BASELINE config 4)",
           "scale": scale, "vertices": int(A.nrows), "edges": int(A.nvals), "value": round(tot / dt, 1), "unit": "TEPS",
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "build_seconds": round(t_build, 2)}
    for p in plans:
        p.free()
    if not args.no_roofline:
        out["spmv_full_pass"] = spmv_pass(ctx, engine, At, scale, iters=12)
    At.free()
    A.free()
    return out


def pmc_child(args):
    """Reduced replay of the bench workloads for the rocprofv3 --pmc passes (live_pmc): no timing, no JSON line."""
    from falkordb_amd import engine
    ctx = engine.Context(0)
    scale = args.scale or 22
    A = ctx.mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
    At = A.transpose()
    roots = pick_roots(A, 8)
    plan = engine.BfsPlan(ctx, A, At)
    plan.run(roots[0], -1, False)
    ctx.set_option("bfs_prof_split", 1)          # name push / pull launches for the profiler
    plan.profile(True)
    for r in roots:
        plan.run(r, -1, False)
    plan.profile(False)
    At.build_tiles()
    engine.bench_spmv(ctx, At, which=2, iters=4)
    plan.free(); At.free(); A.free()
    if not args.no_khop:
        K, dp, dm, _ = khop_inputs(ctx, args.khop_scale, args.edge_factor)
        srcs = p_label_sources(K.nrows)
        for i in range(3):
            b = srcs[i * 1024:(i + 1) * 1024]
            engine.expand_count(ctx, b, [K] * 3)
            if i:
                engine.expand_count(ctx, b, [K] * 3, [dp] * 3, [dm] * 3)
                m_, _ = engine.expand_mat(ctx, b, [K] * 2)          # the emitting path (bp_rows_kernel count / emit)
                m_.free()
        Kt = K.transpose()                                           # the blocked full-pass layout (scales past RMAT-22)
        Kt.build_tiles()
        engine.bench_spmv(ctx, Kt, which=2, iters=3)
        Kt.free()
    ctx.sync()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scale", type=int, default=0, help="R-MAT scale (default: 22 on one GPU, 26 on several)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--alpha", type=float, default=0.0, help="push->pull switch factor (0 = library default)")
    ap.add_argument("--force-dir", type=int, default=0, help="0 auto, 1 push only, 2 pull only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path even with one rank")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (fgpu_set_option)")
    ap.add_argument("--no-khop", action="store_true", help="skip the k-hop MATCH leg (BASELINE config 3)")
    ap.add_argument("--khop-scale", type=int, default=24, help="scale of the k-hop leg that carries the kernel table / PMC traffic")
    ap.add_argument("--khop-batches", type=int, default=32, help="1024-source batches of the :P set to time (0 = all)")
    ap.add_argument("--khop-extra-scales", default="22,26", help="further scales of the same leg (BASELINE metric: 22 / 26), fewer batches")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle checks of the k-hop legs")
    ap.add_argument("--no-varlen", action="store_true", help="skip the config-5 stand-in leg ([*1..4] reach sets, sources sharded over the ranks)")
    ap.add_argument("--no-scale-base", action="store_true", help="skip the RMAT-26 single-GPU base point of the scaling curve")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (traffic = committed / null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

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

    # N = 1: BASELINE config 2 (RMAT-22).  N > 1: BASELINE config 4 / north_star's scaling curve — RMAT-26, the SAME
    # graph at every N (strong scaling); `--gpus 1 --scale 26` gives the curve's base point on one device.
    scale = args.scale or (26 if world > 1 else 22)
    seed = 0x5EED1234 + scale
    ctx = engine.Context(local_rank)
    info = ctx.device_info()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    # ---- synthetic input, resident in HBM before anything is timed ----------------------
    t_build = time.time()
    A_full = ctx.mat_rmat(scale, args.edge_factor, seed)
    n, nnz = A_full.nrows, A_full.nvals
    roots = pick_roots(A_full, 64)
    splits, slab_nnz = None, None
    if use_dist:
        # column-slab partition, boundaries balanced by nnz (prefix sum of in-degrees, multiples of 4096): rank r owns
        # destinations [splits[r], splits[r+1]) and holds A[:, slab] (push) and A'[slab, :] (pull)
        splits = A_full.balanced_splits(world)
        A = A_full.col_slab(int(splits[rank]), int(min(splits[rank + 1], n)))
        A_host = None
        if rank == 0 and not args.no_cpu_baseline:
            A_host = A_full.export_csr()[:2]       # the CPU baseline needs the whole graph; taken before it is freed
        A_full.free()
        At = A.transpose()
        # the communicator lives inside libfgpu.so (fgpu_comm_*): the launcher's channel only carries the unique id
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        td.broadcast_object_list(uid, src=0, device=dev)
        ctx.comm_init_rank(world, rank, uid[0])
        plan = engine.BfsPlan(ctx, A, At, rank, world, splits=splits)
        t = torch.zeros(world, dtype=torch.int64, device=dev)
        t[rank] = A.nvals
        td.all_reduce(t, op=td.ReduceOp.SUM)
        slab_nnz = [int(x) for x in t.tolist()]
    else:
        A = A_full
        At = A.transpose()
        plan = engine.BfsPlan(ctx, A, At)
    plan.tune(alpha=args.alpha, force_direction=args.force_dir)
    ctx.sync()
    t_build = time.time() - t_build

    def run_one(src):
        if use_dist:
            engine.bfs_dist_run([plan], src, -1, False)   # level loop + frontier exchange inside the library
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

    # ---- multi-rank: time split of the timed searches (HIP events inside fgpu_bfs_dist_run), rank 0's view ---
    roofline = None
    spmv = None
    dist_split = None
    if use_dist:
        lm = cm = 0.0
        nl = sc = rl = 0
        ctx.set_option("dist_timing", 1)                     # event pairs per level: only in this untimed replay
        for i in range(min(args.steps, len(roots))):         # untimed replay: per-search event sums + scan counters
            run_one(roots[i])
            a_, b_, c_ = plan.dist_times()
            st = plan.stats()
            lm += a_; cm += b_; nl += c_
            sc += st["scanned_push"] + st["scanned_pull"]
            rl += st["reached"]
        ctx.set_option("dist_timing", 0)
        k = min(args.steps, len(roots))
        nw_bytes = ((n + 4095) // 4096 * 4096) // 8
        alg = 4 * sc + 2 * nw_bytes * nl + 20 * rl           # column ids examined + both bitmaps per level + level/deg of owned discoveries
        tt = torch.tensor([lm, cm, float(alg), float(nl)], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        td.all_gather(gathered, tt)
        per_rank = [[float(x) for x in g.tolist()] for g in gathered]
        if rank == 0:
            ach = alg / max(lm, 1e-9) / 1e6
            roofline = {"bound": "hbm", "kernel": "bfs_fused_kernel (slab mode: every BFS level of rank 0's column slab)",
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None, "alg_bytes_per_launch": int(alg / max(nl, 1)),
                        "avg_launch_us": round(lm / max(nl, 1) * 1e3, 2), "launches": int(nl),
                        "timing": "HIP events around every level kernel and every frontier exchange inside "
                                  "fgpu_bfs_dist_run, replay of the first roots after the timed region"}
            dist_split = {"searches": k, "per_search_ms": {"level_kernels": round(lm / k, 4), "frontier_exchange": round(cm / k, 4)},
                          "exchange": "all-gather-v of the ranks' owned frontier words into every rank's bitmap: grouped "
                                      "ncclSend / ncclRecv over RCCL (xGMI), inside libfgpu.so; includes the wait for the slowest rank",
                          "per_rank_ms_per_search": [{"rank": r, "level_kernels": round(x[0] / k, 4),
                                                      "frontier_exchange": round(x[1] / k, 4)} for r, x in enumerate(per_rank)],
                          "frontier_bitmap_bytes": int(nw_bytes), "levels_per_search": round(nl / k, 2)}

    if not args.no_roofline and not use_dist:
        plan.profile(True)
        for i in range(args.steps):
            plan.run(roots[i % len(roots)], -1, False)
        prof = plan.profile_read()
        plan.profile(False)
        steps = [p for p in prof if p["launches"]]
        if steps:
            # The dominant kernel of the workload is bfs_fused_kernel: ONE kernel runs every level, and this pass
            # launches the same instantiation (<.., 0>) as the timed blind level loop, one level at a time with
            # the direction read back from the control block.  `roofline` is that kernel over all its level
            # launches; the split by direction (push levels are latency / atomic bound, pull levels stream
            # column ids) is in `by_direction`.  `traffic` is filled in by the live PMC passes at the end.
            launches = sum(p["launches"] for p in steps)
            tot_ms = sum(p["ms"] for p in steps)
            tot_bytes = sum(p["alg_bytes"] for p in steps)
            per_launch_bytes = tot_bytes / launches
            per_launch_ms = tot_ms / launches
            ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
            for p in steps:
                p["kernel"] = "bfs_fused_kernel " + p["kernel"].split(" ", 2)[-1]    # "(push level)" / "(pull level)"
            roofline = {"bound": "hbm", "kernel": "bfs_fused_kernel (every BFS level: push and pull launches)",
                        "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": None,
                        "alg_bytes_per_launch": int(per_launch_bytes), "avg_launch_us": round(per_launch_ms * 1e3, 2),
                        "launches": int(launches),
                        "timing": "HIP events on the ctx stream around each level launch of a second, level-synchronous "
                                  "pass over the same roots (same kernel instantiation as the timed blind loop)",
                        "by_direction": [{"kernel": p["kernel"], "ms_total": round(p["ms"], 4),
                                          "launches": int(p["launches"]),
                                          "avg_launch_us": round(p["ms"] / p["launches"] * 1e3, 2),
                                          "alg_bytes_per_launch": int(p["alg_bytes"] / p["launches"]),
                                          "traffic": None,
                                          "GBps": round(p["alg_bytes"] / max(p["ms"], 1e-9) / 1e6, 2),
                                          "frac": round(p["alg_bytes"] / max(p["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
                                         for p in steps],
                        "note": "BFS levels are latency / L2-line bound (bitmap probes, atomics); the HBM-bound "
                                "kernel of this path is the full-pass boolean SpMV reported in spmv_full_pass"}
        # the north-star full-matrix boolean SpMV pass (dense frontier, no mask, no early exit)
        # (LDS-tiled layout, tiled.hip) with the CSR pull kernel's figure beside it
        spmv = spmv_pass(ctx, engine, At, scale)

    # ---- CPU baseline on this box's host cores, bounded sample, rank 0 / N=1 only -------------------
    # The reference's path is LAGraph's push/pull BFS over SuiteSparse:GraphBLAS with OpenMP inside every
    # vxm / mxv; neither library exists in this image, so the stand-in is the oracle's OpenMP
    # direction-optimizing BFS (oracle/oracle_omp.c, same algorithm family) on every host core, with the
    # serial queue BFS (oracle/oracle.c) beside it.  Baseline only: the roofline fraction is the quality bar.
    cpu = None
    if not args.no_cpu_baseline and rank == 0:
        import oracle
        rp, ci = A_host if use_dist else A.export_csr()[:2]
        a = oracle.CSR(n, n, rp, ci)
        A_host = None
        at = None
        if not use_dist and scale < 25:          # RMAT-25+ (17 GB of host arrays with the transpose): push-only baseline
            trp, tci, _ = At.export_csr()
            at = oracle.CSR(n, n, trp, tci)
        # thread count = the job's CPU quota (cpu_threads): a team beyond it is throttled and its rate is noise
        threads, ncpu, quota = cpu_threads()
        oracle.bfs_omp(a, at, roots[0], -1, threads=threads)           # warm-up: page faults, thread team
        e_cpu, t_cpu, k = 0, 0.0, 0
        rates = []
        budget = args.cpu_seconds * 0.7
        while t_cpu <= budget or k < 5:                                 # cycle the 64 roots until the budget is spent
            r = roots[k % len(roots)]
            t1 = time.perf_counter()
            _, e = oracle.bfs_omp(a, at, r, -1, threads=threads)
            d1 = time.perf_counter() - t1
            t_cpu += d1
            e_cpu += e
            rates.append(e / d1)
            k += 1
        rates.sort()
        e_ser, t_ser, ks = 0, 0.0, 0
        for r in roots:
            t1 = time.perf_counter()
            _, _, e = oracle.bfs(a, r, -1, want_parent=False)
            t_ser += time.perf_counter() - t1
            e_ser += e
            ks += 1
            if t_ser > args.cpu_seconds * 0.3:
                break
        cpu = {"value": round(rates[len(rates) // 2], 1), "unit": "TEPS", "cores": threads, "kind": "port",
               "sample": f"median over {k} BFS runs cycling the 64 roots of the same RMAT-{scale} graph ({t_cpu:.1f} s; aggregate "
                         f"{e_cpu / t_cpu / 1e9:.2f} GTEPS, quartiles {rates[len(rates) // 4] / 1e9:.2f} / "
                         f"{rates[(3 * len(rates)) // 4] / 1e9:.2f}), OpenMP "
                         f"{'push/pull' if at is not None else 'push-only (no transposed copy on the host at this size)'} BFS "
                         f"(oracle/oracle_omp.c orc_bfs_omp) on {threads} threads = the job's CPU quota; CPU stand-in for "
                         f"LAGraph + SuiteSparse:GraphBLAS, which are absent from this image",
               "reference_libs_probe": probe_reference_libs(),
               "host_cpus_visible": ncpu, "cgroup_cpu_quota": quota,
               "serial": {"value": round(e_ser / t_ser, 1), "cores": 1,
                          "sample": f"{ks} roots, {t_ser:.1f} s, serial queue BFS (oracle/oracle.c orc_bfs)"}}

    # ---- base point of the strong-scaling curve the N > 1 runs measure (RMAT-26 on this one GPU) ----------
    base26 = None
    if world == 1 and not use_dist and scale == 22 and not args.no_scale_base:
        base26 = strong_scaling_base(ctx, engine, args)

    # ---- the fgpu_bfs ABI entry itself (host level[] array) beside the plan API the timed region uses -----------
    bfs_host = None
    if not use_dist and rank == 0 and world == 1:
        k = min(args.steps, 16)
        engine.bfs(ctx, A, At, roots[0], -1, want_parent=False)
        t1 = time.perf_counter()
        e_h = 0
        for i in range(k):
            _, _, e = engine.bfs(ctx, A, At, roots[i % len(roots)], -1, want_parent=False)
            e_h += e
        dth = time.perf_counter() - t1
        bfs_host = {"entry": "fgpu_bfs (level[] returned in a host array, one call per search, nothing pipelined)",
                    "steps": k, "ms_per_step": round(dth / k * 1e3, 4), "TEPS": round(e_h / dth, 1),
                    "note": "the timed region uses the plan API (results stay on the device, two plans pipelined)"}

    # ---- BASELINE config 3 / the metric's "k-hop MATCH, RMAT scale-22/26": k-hop legs -----------------------------
    khop = None
    khop_scales = {}
    khop_emit = None
    if not args.no_khop and not use_dist and rank == 0:
        khop, (KA, Kdp, Kdm, Khost, kbatches) = khop_leg(ctx, engine, args, args.khop_scale, args.khop_batches)
        if not args.no_roofline:
            KAt = KA.transpose()
            khop["spmv_full_pass"] = spmv_pass(ctx, engine, KAt, args.khop_scale, iters=20)
            KAt.free()
        khop_emit = khop_emit_leg(ctx, engine, args, KA, Khost, kbatches)
        Kdp.free(); Kdm.free(); KA.free()
        del Khost
        for sc in [int(x) for x in args.khop_extra_scales.split(",") if x.strip()]:
            # the same leg at the other scales BASELINE's metric names; scale 26: fewer batches, the oracle takes the
            # first 128 rows of batch 0 (its chain for 1024 rows is ~1 minute of CPU at that size)
            leg, (a_, dp_, dm_, _, _) = khop_leg(ctx, engine, args, sc, 8 if sc <= 24 else 4,
                                                 parity_rows=1024 if sc <= 24 else 128)
            dp_.free(); dm_.free(); a_.free()
            leg.pop("kernels", None)
            khop_scales[str(sc)] = leg

    # ---- BASELINE config 5's stand-in: [*1..4] reach sets, sources sharded over the ranks, no collective ----------
    varlen = None
    if not args.no_varlen and (world > 1 or (not use_dist and not args.no_khop)):
        fl_v, dt_v, nb_v, un_v = varlen_leg(ctx, engine, args, rank, world)
        if world > 1:
            tv = torch.tensor([float(fl_v), dt_v, float(nb_v), float(un_v)], dtype=torch.float64, device=dev)
            mx = tv.clone()
            td.all_reduce(tv, op=td.ReduceOp.SUM)
            td.all_reduce(mx, op=td.ReduceOp.MAX)
            fl_v, nb_v, un_v, dt_v = tv[0].item(), tv[2].item(), tv[3].item(), mx[1].item()
        varlen = {"workload": "BASELINE config 5 stand-in (LDBC SF100 is not available offline): R-MAT scale 19, edge factor 38 "
                              "(0.5 M vertices, ~20 M edges), `[*1..4]` reach sets (fgpu_expand_levels: per-hop frontiers + the "
                              "DISTINCT union) from 1024-source batches of label :P, 0.1 % tombstones + pending adds on every hop",
                  "sharding": f"source batches round-robin over {world} rank(s), adjacency replicated, no collective",
                  "batches": int(nb_v), "TEPS": round(fl_v / dt_v, 1), "ms_per_batch_per_rank": round(dt_v / max(nb_v / world, 1) * 1e3, 3),
                  "distinct_pairs": int(un_v), "scaling": "weak"}

    # ---- HBM traffic per launch, measured now: rocprofv3 --pmc passes over a reduced replay -------------
    pmc = None
    if not args.no_pmc and not args.no_roofline and not use_dist and rank == 0:
        pmc = live_pmc(args)

        def hbm(name):
            e = pmc.get(name) if isinstance(pmc, dict) else None
            return e["hbm_bytes_per_dispatch"] if e else None
        src = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py over `--pmc-child` (reduced replay, "
               "same graphs and kernels); FETCH_SIZE x2 (gfx950) + WRITE_SIZE, bytes per launch")
        if "error" in pmc:
            src = f"live PMC passes failed ({pmc['error']}); committed profiles/traffic.json used where its source hash matches"
        if roofline:
            tot, cnt = 0, 0
            for d in roofline["by_direction"]:                     # per direction, then weighted by this run's launches
                d["traffic"] = hbm(d["kernel"])
                if d["traffic"] is not None:
                    tot += d["traffic"] * d["launches"]
                    cnt += d["launches"]
            roofline["traffic"] = int(tot / cnt) if cnt == roofline["launches"] and cnt else \
                committed_traffic("bfs_fused_kernel", scale)
            roofline["traffic_source"] = src
        if spmv:
            spmv["traffic"] = hbm(spmv["kernel"]) or committed_traffic("tiled_mxv_kernel", scale)
            if khop and khop.get("spmv_full_pass"):
                khop["spmv_full_pass"]["traffic"] = hbm(khop["spmv_full_pass"]["kernel"])
        if khop and khop.get("roofline"):
            khop["roofline"]["traffic"] = hbm(khop["roofline"]["kernel"])
            khop["roofline"]["traffic_source"] = src
            khop["pmc"] = {k: v for k, v in pmc.items() if k.startswith("bp_")} if "error" not in pmc else pmc
        if khop_emit:
            for leg in khop_emit.values():
                for kk in leg["kernels"]:
                    kk["traffic"] = hbm(kk["kernel"])

    if rank == 0:
        st0 = stats_by_root[roots[0]]
        # what BASELINE.json's metric names — k-hop MATCH TEPS at RMAT-22 / 24 / 26 and the fraction of the HBM roofline of the
        # kernels on the path — in the keys the driver keeps (`config`, `roofline`); the full objects follow below
        khop_summary = None
        secondary = []
        if khop:
            def brief(leg):
                return {"clean_TEPS": leg["clean"]["TEPS"], "clean_ms_per_batch": leg["clean"]["ms_per_batch"],
                        "dirty_TEPS": leg["dirty"]["TEPS"], "dirty_ms_per_batch": leg["dirty"]["ms_per_batch"],
                        "out_nnz_per_batch": leg["clean"]["out_nnz"] // leg["batches_timed"],
                        "batches": leg["batches_timed"], "parity_checked": bool(leg["parity"].get("checked")),
                        "parity_ok": leg["parity"].get("ok"), "parity_rows": leg["parity"].get("rows"),
                        "cpu_TEPS": (leg.get("cpu_baseline") or {}).get("value")}
            khop_summary = {"metric": "TEPS = traversed edges (sum over hops of flops) / wall time of fgpu_expand_count, 3-hop MATCH, "
                                      "1024-source batches of label :P, RMAT scale -> figures",
                            str(args.khop_scale): brief(khop)}
            for sc, leg in khop_scales.items():
                khop_summary[sc] = brief(leg)
            if khop_emit:
                khop_summary["materialised"] = {k: {"ms_per_batch": v["ms_per_batch"], "TEPS": v["TEPS"],
                                                    "out_nnz_per_batch": v["out_nnz_per_batch"],
                                                    "parity_ok": (v.get("parity") or {}).get("ok")}
                                                for k, v in khop_emit.items()}
            if khop.get("roofline"):
                r_ = khop["roofline"]
                secondary.append({"kernel": r_["kernel"], "workload": f"RMAT-{args.khop_scale} 3-hop MATCH, hop 3 (count)",
                                  "achieved": r_["achieved"], "unit": "GB/s", "frac": r_["frac"], "traffic": r_["traffic"],
                                  "alg_bytes_per_launch": r_["alg_bytes_per_launch"], "avg_launch_us": r_["avg_launch_us"]})
            if khop_emit:
                for nm, leg in khop_emit.items():
                    for kk in leg["kernels"]:
                        if kk["kernel"].startswith("bp_rows_kernel"):
                            secondary.append({"kernel": kk["kernel"], "workload": f"RMAT-{args.khop_scale} {nm} emission (ballot transpose)",
                                              "achieved": kk["GBps"], "unit": "GB/s", "frac": kk["frac"], "traffic": kk.get("traffic"),
                                              "alg_bytes_per_launch": kk["alg_bytes_per_launch"], "avg_launch_us": kk["avg_launch_us"]})
        for sp in (spmv, (khop or {}).get("spmv_full_pass"), (base26 or {}).get("spmv_full_pass")):
            if sp:
                secondary.append({"kernel": sp["kernel"], "workload": f"RMAT-{sp['scale']} full-matrix boolean SpMV pass (north-star case)",
                                  "achieved": sp["achieved"], "unit": "GB/s", "frac": sp["frac"], "frac_warm": sp["warm"]["frac"],
                                  "traffic": sp.get("traffic"), "alg_bytes_per_launch": sp["alg_bytes"],
                                  "avg_launch_us": sp["avg_launch_us"], "cache_state": "cold"})
        if roofline is not None:
            roofline["secondary"] = secondary
        out = {
            "metric": "traversed edges/sec (TEPS) on BFS (boolean vxm frontier loop), synthetic R-MAT",
            "value": round(teps, 1),
            "unit": "TEPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"RMAT scale-{scale} BFS (boolean GrB_vxm frontier loop), edge factor {args.edge_factor}, "
                            f"64 Graph500-style roots, directed, deduplicated",
                "scale": scale, "vertices": int(n), "edges": int(nnz),
                "parallelism": ("1 GPU" if not use_dist else
                                f"{world} column slabs balanced by nnz, one rank per GPU; per level one kernel per rank + one "
                                f"all-gather-v of the frontier bitmap over RCCL/xGMI, loop and collective inside libfgpu.so"),
                "slab_splits": [int(x) for x in splits] if splits is not None else None,
                "slab_nnz": slab_nnz,
                "direction": {0: "auto push/pull", 1: "push only", 2: "pull only"}[args.force_dir],
                "device": info["name"], "build_seconds": round(t_build, 2),
                "root0_levels": st0["levels"], "root0_push_levels": st0["push_levels"],
                "root0_pull_levels": st0["pull_levels"],
                "khop_match": khop_summary,
                "config5_varlen": ({"TEPS": varlen["TEPS"], "batches": varlen["batches"]} if varlen else None),
            },
            "roofline": roofline,
            "time_split": dist_split,
            "spmv_full_pass": spmv,
            "rmat26_single_gpu": base26,
            "bfs_host_arrays": bfs_host,
            "khop_match": khop,
            "khop_match_other_scales": khop_scales or None,
            "khop_materialised": khop_emit,
            "config5_varlen": varlen,
            "cpu_baseline": cpu,
            "pmc": ({k: v for k, v in pmc.items() if not k.startswith("bp_")} if pmc else None),
        }
        try:
            # RCCL writes its version banner through C stdio, which would otherwise be flushed at exit — AFTER the JSON
            # line; flush it first so that the bench line is the last line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
