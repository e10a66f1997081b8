#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: traversed edges/sec (TEPS) on k-hop MATCH over synthetic R-MAT graphs, with the
fraction of the HBM roofline of the dominant kernel.

  python bench.py --gpus N --steps K --warmup W

A "step" is ONE label scan of `MATCH (a:L)-->()-->()-->(c)` through the device core of CondTraverseOp::expand_batch
(cond_traverse.rs:452-751: F = build(sources); F = delta_lmxm(F; hop) for each of the three hops, matrix.rs:1317-1402) as
ONE WHOLE-FRONTIER call: every source of the label (~N/16 vertices: 261 6xx rows at scale 22) goes into a single
fgpu_expand_count (count + order-independent checksum of the (row, dest) result; H2D of the sources and D2H of the sums
included, adjacency and its cached transpose resident in HBM).  SURVEY.md §7 "hard part 1" / §8d define the k-hop measurement
this way — "all active sources of a scan" — and name the reference's 1024-row child batch (batch.rs:81) as the thing NOT to
inherit; rounds 1-5 still drove the engine one 1024-row call at a time (kept as `secondary.khop22.batch1024_*`).  Inside the
call the library drops the rows whose source has no out-edge (half of an R-MAT label), cuts the live rows into passes of 1024
(one 128-byte line of the bit state per vertex) and deals the passes to 3 lanes (streams) of the context (spgemm.hip
expand_count_scan; `config` states sources per call, live rows, pass width, passes and lanes).  The graph is R-MAT scale 22
(edge factor 16, directed, deduplicated; `--scale` changes it); the labels are the 16 residue classes of hash(id) % 16 —
step i scans class i mod 16, class 0 = the label :P of SURVEY.md §8d first — clean layers.  W untimed steps, then exactly K
timed ones between two fences (barrier + torch.cuda.synchronize()); `value` = sum over the K steps (and over the ranks) of
the traversed edges (sum over hops of flops) / max-over-ranks wall time.

READ `value` WITH `roofline.hop.frac`.  "TEPS" here is ALGEBRAIC: the edges the reference's GrB_mxm chain would traverse
(sum over hops of flops, verified against the oracle), not edges this engine touches one by one — a bit-parallel pull over A'
serves ~100 counted edges per physical row gather, so TEPS x 4 bytes exceeds the HBM peak by construction.  How well the
hardware is used is what `roofline` says: the dominant kernel's (and its hop's) compulsory bytes over its measured time.

N > 1: the label scans are dealt round-robin over the ranks (rank r scans classes r, r + N, ...), the adjacency is
replicated, no collective on the data path (k-hop MATCH shards its source rows) — weak scaling.  The
RMAT-26 BFS over column slabs with one RCCL frontier all-gather-v per level (BASELINE config 4) runs as a secondary leg
of the same line (`secondary.bfs26_dist`; its single-GPU base point is `secondary.bfs26` of the N = 1 line).

Rank 0 prints TWO lines: `DETAIL {...}` (every leg in full; also written to bench_detail.json) and then, LAST, the
compact bench line (< 4 KB, asserted): metric / value / config of the headline, `roofline` of its dominant kernel
(xp_stream_kernel: HIP-event average over one scan replayed, compulsory bytes per launch, live PMC traffic; `hop` = the
count hop, stream + fold, against the same bytes), `cpu_baseline` (the oracle's OpenMP Gustavson chain on this box's host
cores, bounded sample, quartiles; `probe` = what was looked for of the reference's own CPU library; `scipy` = a second point),
in-run `parity` against the oracle (first 1024 rows live, the whole :P scan against a committed oracle run), and
`secondary`: numbers only for BFS 22 / 26, boolean SpMV 22 / 24 / 26, k-hop 24 / 26, dirty layers, the materialised form,
the host-array entries and the config-5 stand-in.
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


STALLED_THREADS = []          # legs whose helper threads never came back: main() must not wait for them at exit
DIST_FAILED = []              # the RCCL leg raised on this rank: the other ranks may be stuck in its collectives


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
    ("xp_stream_kernel", r"xp_stream_kernel"),
    ("xp_fold_kernel", r"xp_fold_kernel"),
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
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--scale", str(args.scale or 22)] + \
            (["--no-bfs"] if args.no_bfs else [])
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


def bfs_threads_child(args, scale, threads=3, steps=128, warmup=16, timeout_s=150):
    """The BFS leg from `threads` query threads (each its own lane and pair of plans), run in a CHILD process with a timeout:
    host threads that spin on device flags cannot be cancelled, so a stall there must not take the bench line with it."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--leg", "bfs", "--no-roofline", "--scale", str(scale), "--steps", str(steps),
           "--warmup", str(warmup), "--bfs-threads", str(threads), "--edge-factor", str(args.edge_factor)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        if r.returncode != 0:
            return {"error": f"child exited {r.returncode}: {r.stderr[-200:]}"}
        line = json.loads(r.stdout.strip().splitlines()[-1])
        q = (line.get("query_threads") or [{}])[0]
        return dict(q, single_thread_ms_in_child=line.get("ms_per_step"))
    except subprocess.TimeoutExpired:
        return {"error": f"child exceeded {timeout_s} s"}
    except (OSError, ValueError, IndexError) as e:
        return {"error": repr(e)}


def operator_child(scale=20, timeout_s=200):
    """CondTraverseOp::expand_batch (the drop-in unit: label probes, layer waits, the chain, the two result columns) beside
    the bare fgpu_expand of the same 1024-source 2-hop batch — tools/bench_paths.py host, in a child process with a deadline
    (the host layer loads its graph edge by edge: ~50 s at RMAT-20, which is why this leg is not run at RMAT-22 here;
    profiles/ holds the RMAT-22 run)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_paths.py"), "host", str(scale)], capture_output=True,
                           text=True, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"did not finish within {timeout_s} s"}
    for l in r.stdout.splitlines()[::-1]:
        if l.startswith("{") and "host_expand_batch" in l:
            d = json.loads(l)
            return {"scale": d["scale"], "rows_out": d["rows_out"], "ms_operator": d["ms_cpp_expand_batch"],
                    "ms_bare_fgpu_expand": d["ms_bare_fgpu_expand"],
                    "ratio": round(d["ms_cpp_expand_batch"] / max(d["ms_bare_fgpu_expand"], 1e-9), 3), "graph_load_s": d["graph_load_s"]}
    return {"error": (r.stderr or r.stdout)[-200:]}


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


def p_label_sources(n, cls=0):
    """The synthetic label :P of SURVEY.md §8d: ids with hash(id) % 16 == 0 (hash = splitmix64 finaliser), ascending.
    `cls` = the residue: 16 disjoint labels of the same size (the timed steps of the headline walk them, :P first)."""
    ids = np.arange(n, dtype=np.uint64)
    return ids[mix64_np(ids) % np.uint64(16) == np.uint64(cls)]


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


def khop_leg(ctx, engine, args, scale, nb_want, graph=None, parity_rows=1024, scan_sources=8192):
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
        # kernel table: ONE whole-frontier call over `scan_sources` rows (the form the line quotes: 1024 live rows per pass,
        # 128-byte rows of the bit state) with HIP events around every modelled launch
        lanes_ = ctx.get_option("expand_scan_lanes")
        ctx.set_option("expand_scan_lanes", 1)                  # (one lane: an event interval is then one kernel's time)
        engine.expand_count(ctx, srcs[:max(min(len(srcs), scan_sources), B)], *layers)
        ctx.prof_enable(True)
        t1 = time.perf_counter()
        engine.expand_count(ctx, srcs[:max(min(len(srcs), scan_sources), B)], *layers)
        dt_prof = time.perf_counter() - t1
        prof = ctx.prof_read()
        ctx.prof_enable(False)
        ctx.set_option("expand_scan_lanes", lanes_)
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
                     "ms_scan_call_with_kernel_events": round(dt_prof * 1e3, 3)}
    # ---- the same label as ONE whole-frontier call (the headline's form): `scan_sources` sources, clean and dirty -------
    S = min(len(srcs), scan_sources)
    if S > B:
        sc = {"sources": int(S)}
        for name, layers in (("clean", ([A] * hops, None, None)), ("dirty", ([A] * hops, [dp] * hops, [dm] * hops))):
            engine.expand_count(ctx, srcs[:S], *layers)                 # warm: the worker lanes' pools
            # (a call of a few passes deals them to its lanes as they come free: a lane that got one pass in the warm call and two
            # in the timed one grows its pool — a hipMalloc of 10+ GB at RMAT-26 — inside the timed call; two timed calls, the faster
            # one quoted, both recorded)
            ds = []
            for _ in range(2):
                ctx.sync()
                t1 = time.perf_counter()
                r = engine.expand_count(ctx, srcs[:S], *layers)
                ds.append(time.perf_counter() - t1)
            d = min(ds)
            sc[name] = {"ms_per_call": round(d * 1e3, 3), "TEPS": round(r[2] / d, 1), "flops": int(r[2]), "out_nnz": int(r[0]),
                        "ms_per_1024_sources": round(d * 1e3 / (S / 1024), 4), "ms_calls": [round(x * 1e3, 3) for x in ds]}
            if name == "clean":
                sc["live_sources"] = ctx.get_option("expand_scan_last_live")
                sc["passes"] = ctx.get_option("expand_scan_last_passes")
            # the whole-frontier path on batch 0 alone (forced: the call is below the switch-over size) must give the
            # oracle-checked triple of the 1024-row call, checksum included (row i of the call = row i of the batch)
            smin = ctx.get_option("expand_scan_min")
            try:
                ctx.set_option("expand_scan_min", 1)
                forced = engine.expand_count(ctx, batches[0], *layers)
            finally:
                ctx.set_option("expand_scan_min", smin)
            sc[name]["batch0_through_the_scan_path_ok"] = bool(tuple(forced) == tuple(batch0[name]))
            if tuple(forced) != tuple(batch0[name]):
                raise SystemExit(f"bench.py: scale {scale} {name}: batch 0 through the whole-frontier path {forced} != {batch0[name]}")
        out["scan"] = sc
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
                           "timing": "HIP events around each launch (fgpu_prof_*), clean layers, one whole-frontier call replayed"}
        byname = {k["kernel"]: k for k in kern}
        if d["kernel"] == "xp_stream_kernel" and "xp_fold_kernel" in byname:
            hop_ms = d["ms"] + byname["xp_fold_kernel"]["ms"]
            out["roofline"]["hop_frac"] = round(d["alg_bytes"] / max(hop_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)
            out["roofline"]["hop_us"] = round(hop_ms / d["launches"] * 1e3, 2)
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
        # the WHOLE timed batch 0 against the committed oracle run of exactly these inputs (tests/golden/make_khop26_golden.py:
        # two and a half minutes of CPU at RMAT-26, done once), when one exists for this scale
        gpath = os.path.join(ROOT, "tests", "golden", "khop%d_batch0.json" % scale)
        if parity_rows < B and os.path.exists(gpath) and args.edge_factor == 16:
            gold = json.load(open(gpath))
            full_ok = (gold.get("edges") == int(nnz) and gold.get("rows") == B and
                       all(tuple(batch0[k_]) == (gold[k_]["nnz"], gold[k_]["checksum"], gold[k_]["flops"]) for k_ in ("clean", "dirty")))
            out["parity"]["full_batch_vs_committed_oracle_run"] = {"ok": bool(full_ok), "rows": B, "golden": "tests/golden/khop%d_batch0.json" % scale}
            if not full_ok:
                raise SystemExit(f"bench.py: k-hop parity FAILED at scale {scale} against {gpath}: gpu {batch0}")
            out["parity"]["rows"] = B
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
        # the host-array entry (fgpu_expand: pinned result blocks filled by DMA, ids widened on the device): the first call pays
        # the pinning of its result blocks, later ones reuse them (fgpu_free returns them to the context's pool)
        t_first_host, t_host, t_host64 = None, [], []
        for rep in range(2):                       # pass 0 sizes the pool (results of a batch differ by tens of per cent), pass 1 is timed
            for b in bl[:5]:
                t1 = time.perf_counter()
                r_ = engine.expand32(ctx, b, [A] * hops)       # the device's own 32-bit arrays, two DMAs (what the operator takes)
                d1 = time.perf_counter() - t1
                if rep == 0 and b is bl[0]:
                    t_first_host = d1
                    rp, dest = r_[0].astype(np.uint64), r_[1].astype(np.uint64)
                if rep == 1:
                    t_host.append(d1)
                del r_
                t1 = time.perf_counter()
                r_ = engine.expand(ctx, b, [A] * hops)         # GrB_Index-wide arrays: ids widened on the device, twice the PCIe bytes
                d1 = time.perf_counter() - t1
                if rep == 0 and b is bl[0] and not (np.array_equal(r_[0], rp) and np.array_equal(r_[1], dest)):
                    raise SystemExit(f"bench.py: fgpu_expand and fgpu_expand32 disagree ({name})")
                if rep == 1:
                    t_host64.append(d1)
                del r_
        # the streamed form (fgpu_expand_stream_*): 64-row chunks, the caller walking each chunk (here: touching its ends)
        t1 = time.perf_counter()
        st_ = engine.ExpandStream(ctx, bl[0], [A] * hops, chunk_rows=64, dest_bits=64)
        t_first = None
        seen = 0
        for _, rp_c, d_c in st_:
            if t_first is None:
                t_first = time.perf_counter() - t1
            seen += len(d_c)
        t_stream = time.perf_counter() - t1
        st_.close()
        assert seen == len(dest)
        rec = {"hops": hops, "rows": rows, "batches": nb, "ms_per_batch": round(dt / nb * 1e3, 3),
               "TEPS": round(tot_f / dt, 1), "out_nnz_per_batch": int(tot_n // nb),
               "host_arrays_ms_first_batch": round(t_first_host * 1e3, 3),
               "host_arrays_ms": round(sorted(t_host)[len(t_host) // 2] * 1e3, 3),
               "host_arrays_u64_ms": round(sorted(t_host64)[len(t_host64) // 2] * 1e3, 3),
               "stream_ms": {"first_chunk": round(t_first * 1e3, 3), "all_chunks": round(t_stream * 1e3, 3), "chunk_rows": 64},
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


def types_ns(**kw):
    import types
    return types.SimpleNamespace(**kw)


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
    """Reduced replay of the bench workloads for the rocprofv3 --pmc passes (live_pmc): no timing, no JSON line.  One
    graph (the headline scale): three k-hop batches clean, two dirty, the emitting path, then BFS searches with the
    direction-named instantiations and the full-pass SpMV."""
    from falkordb_amd import engine
    ctx = engine.Context(0)
    scale = args.scale or 22
    A, dp, dm, _ = khop_inputs(ctx, scale, args.edge_factor)
    srcs = p_label_sources(A.nrows)
    # the k-hop kernels in the form the line quotes: whole-frontier calls (passes of 1024 live rows, 128-byte rows of the state)
    S = 6144 if scale <= 24 else 4096
    ctx.set_option("expand_scan_lanes", 1)                     # (one stream: the counters are read per dispatch)
    engine.expand_count(ctx, srcs[:S], [A] * 3)
    engine.expand_count(ctx, srcs[:S], [A] * 3, [dp] * 3, [dm] * 3)
    for i in range(1, 3):
        b = srcs[i * 1024:(i + 1) * 1024]
        m_, _ = engine.expand_mat(ctx, b, [A] * 2)              # the emitting path (bp_rows_kernel count / emit)
        m_.free()
    if not args.no_bfs:
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
        plan.free(); At.free()
    dp.free(); dm.free(); A.free()
    ctx.sync()
    ctx.close()


def _kernel_rows(prof, top=10):
    return [{"kernel": k["kernel"], "ms_total": round(k["ms"], 3), "launches": k["launches"],
             "avg_launch_us": round(k["ms"] / max(k["launches"], 1) * 1e3, 2),
             "alg_bytes_per_launch": int(k["alg_bytes"] / max(k["launches"], 1)),
             "GBps": round(k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6, 1),
             "frac": round(k["alg_bytes"] / max(k["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
            for k in sorted(prof, key=lambda k: -k["ms"])[:top]]


def khop_headline(ctx, engine, args, scale, rank, world, fence, reduce_max, reduce_sum):
    """THE timed region of the bench line (module docstring): K WHOLE-FRONTIER calls of the 3-hop MATCH — one
    fgpu_expand_count per label scan, all ~N/16 sources of the label in the call, clean layers, count + checksum — W untimed
    calls before.  Returns (line fields, detail, graph tuple)."""
    hops, B = 3, 1024
    t0 = time.time()
    want_host = rank == 0 and not args.no_parity
    A, dp, dm, host = khop_inputs(ctx, scale, args.edge_factor, want_host=want_host)
    ctx.sync()
    t_build = time.time() - t0
    n, nnz = A.nrows, A.nvals
    srcs = p_label_sources(n)                       # label :P (class 0)
    nb_all = len(srcs) // B
    labels = {0: srcs}

    def scan(i):                                    # this rank's i-th label scan: classes round-robin over the ranks, :P first
        c = (rank + i * world) % 16
        if c not in labels:
            labels[c] = p_label_sources(n, c)
        return labels[c][:args.sources_per_call] if args.sources_per_call > 0 else labels[c]
    clean = ([A] * hops, None, None)
    dirty = ([A] * hops, [dp] * hops, [dm] * hops)
    # snapshot preparation (untimed, once per matrix version like the upload itself): the cached transpose of A, the pull
    # item lists and the device pools are built by the first call that needs them
    t_prep = time.perf_counter()
    batch0 = srcs[:B]
    first = engine.expand_count(ctx, batch0, *clean)      # (also the reference's own unit: one 1024-row child batch)
    ctx.sync()
    t_prep = time.perf_counter() - t_prep                # transpose of A, pull item lists, the partitioned layout of the count hop, pools
    for i in range(max(args.warmup, 1)):                 # (>= 1: the worker lanes' pools are grown by the first whole-frontier call)
        engine.expand_count(ctx, scan(i), *clean)
    timed = [scan(i) for i in range(args.steps)]
    fence()
    t1 = time.perf_counter()
    tot_f = tot_n = 0
    cs = 0
    scan0 = None
    for b in timed:
        nn, c, f = engine.expand_count(ctx, b, *clean)
        if scan0 is None:
            scan0 = (nn, c, f)
        tot_n += nn
        tot_f += f
        cs = (cs + c) & 0xFFFFFFFFFFFFFFFF
    fence()
    dt = reduce_max(time.perf_counter() - t1)
    flops_all = reduce_sum(tot_f)
    nnz_all = reduce_sum(tot_n)
    line = {"value": round(flops_all / dt, 1), "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4)}
    scan_cfg = {"sources_per_call": int(len(timed[0])) if timed else int(len(srcs)),
                "live_sources_last_call": ctx.get_option("expand_scan_last_live"), "passes_last_call": ctx.get_option("expand_scan_last_passes"),
                "pass_rows": ctx.get_option("expand_scan_rows"), "lanes": ctx.get_option("expand_scan_lanes")}
    # what a NEW matrix version costs once the pools are warm (t_prep above also grew them): a second snapshot of the same
    # matrix (transposed twice on the device: no cache is shared with A), its first batch against a steady one
    snap_prep = None
    if rank == 0 and scale <= 24:
        try:
            t_ = time.perf_counter()
            engine.expand_count(ctx, batch0, *clean)
            ctx.sync()
            steady = time.perf_counter() - t_
            At = A.transpose()
            A2 = At.transpose()
            At.free()
            ctx.sync()
            t_ = time.perf_counter()
            r2 = engine.expand_count(ctx, batch0, [A2] * hops, None, None)
            ctx.sync()
            t_ = time.perf_counter() - t_
            if r2 == first:
                snap_prep = round(max(t_ - steady, 0.0) * 1e3, 2)
            A2.free()
        except Exception as e:                        # a report field, never the reason the bench fails
            snap_prep = None
    det = {"workload": f"RMAT scale-{scale} {hops}-hop MATCH (a:P)-->()-->()-->(c), CondTraverse expand_batch core (masked "
                       f"GrB_mxm ANY_PAIR chain): ONE fgpu_expand_count call per label scan — every source of the label "
                       f"(hash(id) % 16 == class; :P = class 0 first), live rows filtered on the device, passes of "
                       f"{scan_cfg['pass_rows']} live rows on {scan_cfg['lanes']} lanes; clean layers",
           "scale": scale, "vertices": int(n), "edges": int(nnz), "hops": hops, "scan": scan_cfg,
           "label_P_sources": int(len(srcs)), "build_seconds": round(t_build, 2),
           "prep_ms": round(t_prep * 1e3, 2), "snapshot_prep_ms": snap_prep,
           "nnz_dp": int(dp.nvals), "nnz_dm": int(dm.nvals),
           "timed": {"steps": args.steps, "warmup": args.warmup, "seconds": round(dt, 5), "flops": int(flops_all),
                     "out_nnz": int(nnz_all), "rank0_checksum": f"{cs:016x}", "TEPS": line["value"],
                     "step0": {"nnz": int(scan0[0]), "checksum": f"{scan0[1]:016x}", "flops": int(scan0[2])} if scan0 else None}}
    if rank != 0:
        return line, det, (A, dp, dm, host, batch0, None)
    # ---- untimed legs over the same label: the calls it replaces, count only, dirty layers, lanes, kernel table -------
    P = srcs[:args.sources_per_call] if args.sources_per_call > 0 else srcs

    def run(src, layers, cs_=True, reps=1):
        engine.expand_count(ctx, src[:4096], *layers, want_checksum=cs_)
        ctx.sync()
        best = None
        for _ in range(reps):
            t = time.perf_counter()
            r = engine.expand_count(ctx, src, *layers, want_checksum=cs_)
            d = time.perf_counter() - t
            best = d if best is None or d < best else best
        return {"ms_per_call": round(best * 1e3, 3), "TEPS": round(r[2] / best, 1), "sources": int(len(src)),
                "ms_per_1024_sources": round(best * 1e3 / (len(src) / 1024), 4)}, r
    # the reference's own batch shape: one call per 1024-row child batch (batch.rs:81), one thread — what rounds 1-5 timed
    ks = [P[j * B:(j + 1) * B] for j in range(min(nb_all, 32))]
    ctx.sync()
    t = time.perf_counter()
    bn = bf = 0
    for b in ks:
        nn, _, f = engine.expand_count(ctx, b, *clean)
        bn += nn
        bf += f
    d = time.perf_counter() - t
    det["batch_1024"] = {"batches": len(ks), "ms_per_batch": round(d / len(ks) * 1e3, 3), "TEPS": round(bf / d, 1)}
    # ... and the same rows as ONE whole-frontier call: the sums must agree (rows are independent)
    sl, r_sl = run(P[:len(ks) * B], clean)
    det["batch_1024"]["same_rows_one_call"] = dict(sl, agrees=bool((r_sl[0], r_sl[2]) == (bn, bf)))
    if (r_sl[0], r_sl[2]) != (bn, bf):
        raise SystemExit(f"bench.py: whole-frontier call over {len(ks) * B} rows gives (nnz, flops) {(r_sl[0], r_sl[2])}, the "
                         f"1024-row calls over the same rows {(bn, bf)}")
    det["count_only"], _ = run(P, clean, False)
    det["dirty"], _ = run(P, dirty)
    if not args.no_lanes_sweep:
        base_lanes = ctx.get_option("expand_scan_lanes")
        sw = []
        for k in (1, 2, 3, 4):
            ctx.set_option("expand_scan_lanes", k)
            q, r_q = run(P[:32 * B], clean)
            sw.append(dict(q, lanes=k, agrees=bool(len(ks) != 32 or (r_q[0], r_q[2]) == (bn, bf))))
        ctx.set_option("expand_scan_lanes", base_lanes)
        det["lanes_sweep"] = sw
    # every row with a pre-bound destination (multi-hop ExpandInto / CondTraverse with `to` bound): the
    # same batches through fgpu_expand_probe — hops 1-2 as usual, the last hop one bit per row — beside fgpu_expand_count
    try:
        rng_ = np.random.default_rng(5)
        dsts_ = [rng_.integers(0, n, len(b)).astype(np.uint64) for b in ks[:8]]
        engine.expand_probe(ctx, ks[0], dsts_[0], [A] * hops)
        ctx.sync()
        t_ = time.perf_counter()
        hits_ = 0
        for b, d_ in zip(ks[:8], dsts_):
            pr_, _ = engine.expand_probe(ctx, b, d_, [A] * hops)
            hits_ += int(pr_.sum())
        det["pinned_probe"] = {"batches": len(dsts_), "ms_per_batch": round((time.perf_counter() - t_) / len(dsts_) * 1e3, 3),
                               "rows_present": hits_, "what": "fgpu_expand_probe: 1024 (source, bound destination) rows, 3 hops, clean layers"}
    except Exception as e:   # noqa: BLE001 — a secondary figure must not cost the line
        det["pinned_probe"] = {"error": repr(e)[:200]}
    # kernel table of ONE whole-frontier call over 32 K sources on ONE lane (HIP events around every modelled launch: with
    # several lanes the kernels of different passes share the chip and an interval is not one kernel's time; the timed
    # region above runs on `lanes` lanes and is what `value` quotes)
    base_lanes = ctx.get_option("expand_scan_lanes")
    ctx.set_option("expand_scan_lanes", 1)
    engine.expand_count(ctx, P[:4 * B], *clean)
    ctx.prof_enable(True)
    engine.expand_count(ctx, P[:32 * B], *clean)
    prof = ctx.prof_read()
    ctx.prof_enable(False)
    ctx.set_option("expand_scan_lanes", base_lanes)
    det["kernels"] = _kernel_rows(prof)
    roofline = None
    kern = sorted(prof, key=lambda k: -k["ms"])
    if kern:
        d = kern[0]
        ach = d["alg_bytes"] / max(d["ms"], 1e-9) / 1e6
        roofline = {"bound": "hbm", "kernel": d["kernel"], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "alg_bytes_per_launch": int(d["alg_bytes"] / d["launches"]),
                    "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2), "launches": int(d["launches"]),
                    "share_of_kernel_time": round(d["ms"] / max(sum(k["ms"] for k in kern), 1e-9), 3),
                    "timing": "HIP events per launch, one 32 K-source call replayed on one lane"}
        # the HOP the kernel belongs to: the partitioned count hop is stream + fold, and only the stream kernel's bytes are
        # bytes the problem needs (the partial rows between the two are an artefact of the partition)
        byname = {k["kernel"]: k for k in prof}
        if d["kernel"] == "xp_stream_kernel" and "xp_fold_kernel" in byname:
            fo = byname["xp_fold_kernel"]
            hop_ms = d["ms"] + fo["ms"]
            roofline["hop"] = {"kernels": "xp_stream_kernel + xp_fold_kernel", "us_per_pass": round(hop_ms / d["launches"] * 1e3, 2),
                               "achieved": round(d["alg_bytes"] / max(hop_ms, 1e-9) / 1e6, 2),
                               "frac": round(d["alg_bytes"] / max(hop_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}
    lv = engine.expand_levels(ctx, batch0, *clean)
    det["batch0_hop_nnz"] = [int(x) for x in lv["hop_nnz"]]
    # the NON-pushdown form of the headline's own unit: the first 1024 :P rows, 3 hops, every (row_i, dest) pair
    # EMITTED as cond_traverse.rs:644-751 emits them — the CSR left on the device (fgpu_expand_mat) and brought to host arrays
    # (fgpu_expand32); its entry count must equal the count form's
    try:
        if scale > 22:
            raise MemoryError("skipped above scale 22: the result of 1024 rows passes 2^32 entries")
        m_, _ = engine.expand_mat(ctx, batch0, *clean)
        n_emit = int(m_.nvals)
        m_.free()
        ctx.sync()
        t_ = time.perf_counter()
        for _ in range(3):
            m_, f_ = engine.expand_mat(ctx, batch0, *clean)
            m_.free()
        ctx.sync()
        d_dev = (time.perf_counter() - t_) / 3
        r_ = engine.expand32(ctx, batch0, *clean)
        del r_
        t_ = time.perf_counter()
        r_ = engine.expand32(ctx, batch0, *clean)
        d_host = time.perf_counter() - t_
        ok_ = n_emit == first[0] and len(r_[1]) == first[0]
        del r_
        det["emitting_three_hop_1024_rows"] = {"entries": n_emit, "ms_device": round(d_dev * 1e3, 3), "ms_host_arrays": round(d_host * 1e3, 3),
                                               "TEPS_device": round(first[2] / d_dev, 1), "entries_match_count_form": bool(ok_)}
        if not ok_:
            raise SystemExit(f"bench.py: the emitting 3-hop batch holds {n_emit} entries, the count form {first[0]}")
    except MemoryError as e:
        det["emitting_three_hop_1024_rows"] = {"error": repr(e)[:160]}
    return line, det, (A, dp, dm, host, batch0, (first, roofline, scan0))


def khop_parity_and_cpu(ctx, engine, args, A, dp, dm, host, batch0, gpu_first, scale):
    """In-run parity of the first 1024 :P rows (one child batch of the reference, batch.rs:81; clean and dirty) against the
    oracle's delta_lmxm chain, and the CPU baseline from that same oracle run: per 64-row chunk rates give the quartiles.
    BASELINE.md §3 step 1: the box is probed for the reference's own CPU library first (`probe`), and scipy.sparse's CSR
    product of the first 64 rows is timed as the second point it promises."""
    import oracle
    hops = 3
    a, hdp, hdm = host
    threads, ncpu, quota = cpu_threads()
    rows = batch0
    chunks = []
    t1 = time.perf_counter()
    ref_clean = oracle.expand_summary_omp(rows, [(a, None, None)] * hops, chunk=64, threads=threads, per_chunk=chunks)
    t_clean = time.perf_counter() - t1
    t1 = time.perf_counter()
    ref_dirty = oracle.expand_summary_omp(rows, [(a, hdp, hdm)] * hops, chunk=64, threads=threads)
    t_dirty = time.perf_counter() - t1
    gpu_dirty = engine.expand_count(ctx, rows, [A] * hops, [dp] * hops, [dm] * hops)
    ok = tuple(gpu_first) == tuple(ref_clean[:3]) and tuple(gpu_dirty) == tuple(ref_dirty[:3])
    parity = {"checked": True, "ok": bool(ok), "rows": int(len(rows)),
              "what": "(nnz, checksum, flops) of the first 1024 :P rows, clean and dirty layers, fgpu_expand_count vs the oracle's "
                      "delta_lmxm chain (oracle.expand_summary_omp; matrix.rs:1317-1402)",
              "clean": {"nnz": int(ref_clean[0]), "checksum": f"{ref_clean[1]:016x}", "flops": int(ref_clean[2])},
              "dirty": {"nnz": int(ref_dirty[0]), "checksum": f"{ref_dirty[1]:016x}", "flops": int(ref_dirty[2])}}
    if not ok:
        raise SystemExit(f"bench.py: k-hop parity FAILED at scale {scale}: gpu {gpu_first} / {gpu_dirty} vs oracle clean "
                         f"{ref_clean[:3]} dirty {ref_dirty[:3]}")
    rates = sorted(f / max(t, 1e-9) for f, t in chunks if f)
    q = [rates[len(rates) // 4], rates[len(rates) // 2], rates[(3 * len(rates)) // 4]] if rates else [0, 0, 0]
    cpu = {"value": round(ref_clean[2] / t_clean, 1), "unit": "TEPS", "cores": threads, "kind": "port",
           "quartiles": [round(x, 1) for x in q],
           "sample": f"first 1024 :P sources, 3 hops, clean: {t_clean:.1f} s of OpenMP Gustavson ANY_PAIR products (oracle/oracle_omp.c) on "
                     f"{threads} threads (the job's CPU quota); stand-in for SuiteSparse:GraphBLAS, absent here",
           "dirty_TEPS": round(ref_dirty[2] / t_dirty, 1)}
    # what was looked for before settling for the port (graphblas.sh:71-72 builds SuiteSparse:GraphBLAS v10.5.0 from a clone)
    found = probe_reference_libs()
    cpu["probe"] = {"looked_for": "libgraphblas, liblagraph(x), GraphBLAS.h, python-graphblas",
                    "found": {k_: v_ for k_, v_ in found.items() if v_} or None}
    try:                                             # second point: scipy.sparse CSR x CSR (SMMP, one thread), pattern re-binarised per hop
        import scipy.sparse as sp
        k_ = 64
        a_sp = sp.csr_matrix((np.ones(a.nnz, dtype=np.int32), a.colidx.astype(np.int64), a.rowptr.astype(np.int64)), shape=(a.nrows, a.ncols))
        f_sp = sp.csr_matrix((np.ones(k_, dtype=np.int32), (np.arange(k_), rows[:k_].astype(np.int64))), shape=(k_, a.nrows))
        t1 = time.perf_counter()
        for _ in range(hops):
            f_sp = f_sp @ a_sp
            f_sp.data[:] = 1
        t_sp = time.perf_counter() - t1
        fl_sp = chunks[0][0] if chunks else 0
        cpu["scipy"] = {"value": round(fl_sp / t_sp, 1), "unit": "TEPS", "cores": 1, "rows": k_}
    except Exception as e:   # noqa: BLE001 — a report field
        cpu["scipy"] = {"error": repr(e)[:120]}
    return parity, cpu


def bfs_single_leg(ctx, engine, args, scale, A=None, steps=64, warmup=8, want_prof=True, want_spmv=True):
    """BASELINE config 2 (RMAT-22) / the base point of config 4 (RMAT-26): whole BFS searches (the boolean GrB_vxm frontier
    loop behind algo.BFS) from the 64 Graph500-style roots, two plans pipelined, results on the device.  TEPS = sum of
    out-degrees of reached vertices / wall time; per-direction kernel table from a level-synchronous replay."""
    t0 = time.time()
    own = A is None
    if own:
        A = ctx.mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
    At = A.transpose()
    roots = pick_roots(A, 64)
    plans = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
    for p in plans:
        p.tune(alpha=args.alpha, force_direction=args.force_dir)
    ctx.sync()
    t_build = time.time() - t0
    edges, st0 = {}, None
    for r in roots:
        plans[0].run(r, -1, False)
        st = plans[0].stats()
        edges[r] = st["edges_traversed"]
        st0 = st0 or st

    def pipelined(srcs):
        for i, src in enumerate(srcs):
            plans[i % 2].run_async(src, -1, False, 0)   # levels=0: one more than this plan's previous search took
            if i > 0:
                plans[(i - 1) % 2].wait()
        if srcs:
            plans[(len(srcs) - 1) % 2].wait()
    pipelined([roots[i % len(roots)] for i in range(warmup)])
    ctx.sync()
    t1 = time.perf_counter()
    pipelined([roots[i % len(roots)] for i in range(steps)])
    ctx.sync()
    dt = time.perf_counter() - t1
    tot = sum(edges[roots[i % len(roots)]] for i in range(steps))
    out = {"workload": f"RMAT scale-{scale} BFS (boolean GrB_vxm frontier loop), 64 Graph500-style roots, plan API (results on "
                       f"the device, two plans pipelined)",
           "scale": scale, "vertices": int(A.nrows), "edges": int(A.nvals), "TEPS": round(tot / dt, 1), "steps": steps,
           "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "build_seconds": round(t_build, 2),
           "root0": {k: st0[k] for k in ("levels", "push_levels", "pull_levels")}}
    if getattr(args, "bfs_threads", 0) > 1:
        # the same searches from two query threads, each with its own pair of plans on its own lane (stream): one search's
        # light levels — a few microseconds of work under a ~10 us level floor — run under the other's heavy ones
        import threading

        def threads_run(k):
            gate = threading.Barrier(k + 1)
            errs = []

            def work(t):
                try:
                    ps = [engine.BfsPlan(ctx, A, At), engine.BfsPlan(ctx, A, At)]
                    for p_ in ps:
                        p_.tune(alpha=args.alpha, force_direction=args.force_dir)

                    def pipe(srcs):
                        for i, src in enumerate(srcs):
                            ps[i % 2].run_async(src, -1, False, 0)
                            if i > 0:
                                ps[(i - 1) % 2].wait()
                        if srcs:
                            ps[(len(srcs) - 1) % 2].wait()
                    pipe([roots[i % len(roots)] for i in range(t, warmup * k, k)])
                    ctx.sync()
                    gate.wait()
                    pipe([roots[i % len(roots)] for i in range(t, steps, k)])
                    ctx.sync()
                    gate.wait()
                    for p_ in ps:
                        p_.free()
                except BaseException as e:   # noqa: BLE001
                    errs.append(repr(e))
                    gate.abort()
            ths = [threading.Thread(target=work, args=(t,)) for t in range(k)]
            for th in ths:
                th.start()
            try:
                gate.wait()
                t2 = time.perf_counter()
                gate.wait()
                dt2 = time.perf_counter() - t2
            except threading.BrokenBarrierError:
                dt2 = None
            for th in ths:
                th.join()
            if dt2 and not errs:
                return {"threads": k, "ms_per_step": round(dt2 / steps * 1e3, 4), "TEPS": round(tot / dt2, 1), "steps": steps}
            return {"threads": k, "error": errs[:1]}
        out["query_threads"] = [threads_run(args.bfs_threads)]
    if want_prof:
        plan = plans[0]
        plan.profile(True)
        for i in range(steps):
            plan.run(roots[i % len(roots)], -1, False)
        prof = [p for p in plan.profile_read() if p["launches"]]
        plan.profile(False)
        if prof:
            launches = sum(p["launches"] for p in prof)
            tot_ms = sum(p["ms"] for p in prof)
            tot_b = sum(p["alg_bytes"] for p in prof)
            for p in prof:
                p["kernel"] = "bfs_fused_kernel " + p["kernel"].split(" ", 2)[-1]    # "(push level)" / "(pull level)"
            out["roofline"] = {"kernel": "bfs_fused_kernel (every BFS level: push and pull launches)",
                               "achieved": round(tot_b / tot_ms / 1e6, 2), "unit": "GB/s",
                               "frac": round(tot_b / tot_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None,
                               "alg_bytes_per_launch": int(tot_b / launches), "avg_launch_us": round(tot_ms / launches * 1e3, 2),
                               "launches": int(launches), "levels_per_search": round(launches / steps, 2),
                               # the replay is level-synchronous (an event pair per level): its per-launch time exceeds what a
                               # level costs inside the blind loop of the timed region, whose upper bound is wall time / levels
                               "blind_loop_us_per_level_upper_bound": round(dt / steps * 1e6 / max(launches / steps, 1e-9), 2),
                               "timing": "HIP events around each level launch of a level-synchronous replay of the same "
                                         "roots (same kernel instantiation as the timed blind loop)",
                               "by_direction": [dict(r, traffic=None) for r in _kernel_rows(prof)]}
    # the fgpu_bfs ABI entry itself (host level[] array, one call per search) beside the plan API: level[] in a pinned block
    # of the context's pool (fgpu_host_alloc: one DMA), and in pageable numpy memory (staging ring + host copy)
    k = min(steps, 16)
    level_pin = ctx.host_array(A.nrows, np.int32)
    out["host_arrays"] = {"entry": "fgpu_bfs (level[] returned in a host array, one call per search, nothing pipelined)", "steps": k}
    for name, buf in (("pinned", level_pin), ("pageable", np.zeros(A.nrows, dtype=np.int32))):
        engine.bfs(ctx, A, At, roots[0], -1, want_parent=False, level_out=buf)
        t1 = time.perf_counter()
        e_h = 0
        for i in range(k):
            _, _, e = engine.bfs(ctx, A, At, roots[i % len(roots)], -1, want_parent=False, level_out=buf)
            e_h += e
        dth = time.perf_counter() - t1
        out["host_arrays"][name] = {"ms_per_step": round(dth / k * 1e3, 4), "TEPS": round(e_h / dth, 1)}
    out["host_arrays"]["ms_per_step"] = out["host_arrays"]["pinned"]["ms_per_step"]
    del level_pin
    for p in plans:
        p.free()
    if want_spmv and not args.no_roofline:
        out["spmv_full_pass"] = spmv_pass(ctx, engine, At, scale, iters=50 if scale <= 22 else 12)
    return out, A, At, roots, own


def cpu_bfs_baseline(args, A, At, roots, scale, seconds):
    """The oracle's OpenMP direction-optimising BFS on this box's host cores (CPU stand-in for LAGraph + SuiteSparse), a
    bounded sample cycling the 64 roots; median + quartiles."""
    import oracle
    n = A.nrows
    a = oracle.CSR(n, n, *A.export_csr()[:2])
    at = oracle.CSR(n, n, *At.export_csr()[:2]) if At is not None and scale < 25 else None
    threads, ncpu, quota = cpu_threads()
    oracle.bfs_omp(a, at, roots[0], -1, threads=threads)
    rates, t_cpu, k = [], 0.0, 0
    while t_cpu <= seconds or k < 5:
        r = roots[k % len(roots)]
        t1 = time.perf_counter()
        _, e = oracle.bfs_omp(a, at, r, -1, threads=threads)
        d1 = time.perf_counter() - t1
        t_cpu += d1
        rates.append(e / d1)
        k += 1
    rates.sort()
    return {"value": round(rates[len(rates) // 2], 1), "unit": "TEPS", "cores": threads, "kind": "port",
            "quartiles": [round(rates[len(rates) // 4], 1), round(rates[(3 * len(rates)) // 4], 1)],
            "sample": f"median over {k} searches cycling the 64 roots of RMAT-{scale} ({t_cpu:.1f} s), OpenMP push/pull BFS "
                      f"(oracle/oracle_omp.c) on {threads} threads", "host_cpus_visible": ncpu, "cgroup_cpu_quota": quota}


def bfs_dist_leg(ctx, engine, args, scale, rank, world, dev, td, torch, steps, warmup):
    """BASELINE config 4: RMAT-<scale> BFS over a column-slab partition balanced by nnz, one rank per GPU, per level one
    kernel per rank + one all-gather-v of the owned frontier words (grouped ncclSend / ncclRecv over RCCL/xGMI), loop and
    collective inside libfgpu.so (fgpu_bfs_dist_run).  Strong scaling: the same graph at every N."""
    t_build = time.time()
    A_full = ctx.mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
    n, nnz = A_full.nrows, A_full.nvals
    roots = pick_roots(A_full, 64)
    splits = A_full.balanced_splits(world)
    A = A_full.col_slab(int(splits[rank]), int(min(splits[rank + 1], n)))
    A_full.free()
    At = A.transpose()
    uid = [ctx.comm_unique_id() if rank == 0 else None]
    td.broadcast_object_list(uid, src=0, device=dev)
    ctx.comm_init_rank(world, rank, uid[0])              # the communicator lives inside libfgpu.so
    plan = engine.BfsPlan(ctx, A, At, rank, world, splits=splits)
    plan.tune(alpha=args.alpha, force_direction=args.force_dir)
    t = torch.zeros(world, dtype=torch.int64, device=dev)
    t[rank] = A.nvals
    td.all_reduce(t, op=td.ReduceOp.SUM)
    slab_nnz = [int(x) for x in t.tolist()]
    ctx.sync()
    t_build = time.time() - t_build

    def run_one(src):
        engine.bfs_dist_run([plan], src, -1, False)
    edges = {}
    for r in roots:
        run_one(r)
        edges[r] = plan.stats()["edges_traversed"]
    t = torch.tensor([edges[r] for r in roots], dtype=torch.int64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.SUM)                 # slab-local out-degree sums -> global
    edges = {r: int(v) for r, v in zip(roots, t.tolist())}
    for i in range(warmup):
        run_one(roots[i % len(roots)])
    td.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run_one(roots[i % len(roots)])
    td.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    td.all_reduce(tt, op=td.ReduceOp.MAX)
    dt = float(tt.item())
    tot = sum(edges[roots[i % len(roots)]] for i in range(steps))
    # time split of the same searches (HIP event pairs per level inside fgpu_bfs_dist_run), untimed replay
    lm = cm = 0.0
    nl = sc = rl = 0
    ctx.set_option("dist_timing", 1)
    k = min(steps, len(roots))
    for i in range(k):
        run_one(roots[i])
        a_, b_, c_ = plan.dist_times()
        st = plan.stats()
        lm += a_; cm += b_; nl += c_
        sc += st["scanned_push"] + st["scanned_pull"]
        rl += st["reached"]
    ctx.set_option("dist_timing", 0)
    nw_bytes = ((n + 4095) // 4096 * 4096) // 8
    alg = 4 * sc + 2 * nw_bytes * nl + 20 * rl           # column ids examined + both bitmaps per level + level/deg of owned discoveries
    tt = torch.tensor([lm, cm], dtype=torch.float64, device=dev)
    gathered = [torch.zeros_like(tt) for _ in range(world)]
    td.all_gather(gathered, tt)
    per_rank = [[float(x) for x in g.tolist()] for g in gathered]
    out = {"workload": f"RMAT scale-{scale} BFS, adjacency in {world} column slab(s) balanced by nnz, one rank per GPU, one "
                       f"RCCL all-gather-v of the frontier per level inside libfgpu.so",
           "scale": scale, "vertices": int(n), "edges": int(nnz), "ranks": world, "TEPS": round(tot / dt, 1),
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "scaling": "strong",
           "build_seconds": round(t_build, 2), "slab_splits": [int(x) for x in splits], "slab_nnz": slab_nnz,
           "per_search_ms": {"level_kernels": round(lm / k, 4), "frontier_exchange": round(cm / k, 4)},
           "per_rank_ms_per_search": [{"rank": r, "level_kernels": round(x[0] / k, 4), "frontier_exchange": round(x[1] / k, 4)}
                                      for r, x in enumerate(per_rank)],
           "levels_per_search": round(nl / k, 2), "frontier_bitmap_bytes": int(nw_bytes),
           "rank0_level_kernel": {"alg_bytes_per_launch": int(alg / max(nl, 1)), "avg_launch_us": round(lm / max(nl, 1) * 1e3, 2),
                                  "frac": round(alg / max(lm, 1e-9) / 1e6 / HBM_PEAK_GBS, 4)}}
    plan.free(); At.free(); A.free()
    return out


def bfs_gang_leg(engine, args, scale, ndev, torch, steps, warmup, ctx_opts, devices=None):
    """BASELINE config 4 the other way in (dist.hip: "one process, several GPUs" — how a Redis-module process would drive a
    node): rank 0's process opens a context on every device, each owns one column slab, and fgpu_bfs_dist_run drives the gang
    from ONE thread with NO communicator: per level one kernel per device, then every rank's owned frontier words are stored
    straight into every peer's bitmap over xGMI (dist_scatter_kernel, hipDeviceEnablePeerAccess) — no RCCL launch in the loop.
    Strong scaling: the same graph as the one-rank base."""
    t_build = time.time()
    ctxs = [engine.Context(d) for d in (devices if devices is not None else range(ndev))]   # (tests: every rank on device 0)
    try:
        for c in ctxs:
            for kv in ctx_opts:
                k, v = kv.split("=")
                c.set_option(k, int(v))
        full0 = ctxs[0].mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
        n, nnz = full0.nrows, full0.nvals
        roots = pick_roots(full0, 64)
        splits = full0.balanced_splits(ndev)
        plans, keep, slab_nnz = [], [], []
        for r, c in enumerate(ctxs):
            full = full0 if r == 0 else c.mat_rmat(scale, args.edge_factor, 0x5EED1234 + scale)
            A = full.col_slab(int(splits[r]), int(min(splits[r + 1], n)))
            full.free()
            At = A.transpose()
            keep += [A, At]
            slab_nnz.append(int(A.nvals))
            p = engine.BfsPlan(c, A, At, r, ndev, splits=splits)
            p.tune(alpha=args.alpha, force_direction=args.force_dir)
            plans.append(p)
        for c in ctxs:
            c.sync()
        t_build = time.time() - t_build
        edges = {}
        for r in roots:
            engine.bfs_dist_run(plans, r, -1, False)
            edges[r] = sum(p.stats()["edges_traversed"] for p in plans)
        for i in range(warmup):
            engine.bfs_dist_run(plans, roots[i % len(roots)], -1, False)
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            engine.bfs_dist_run(plans, roots[i % len(roots)], -1, False)
        for c in ctxs:
            c.sync()
        dt = time.perf_counter() - t0
        tot = sum(edges[roots[i % len(roots)]] for i in range(steps))
        lm = cm = 0.0
        nl = 0
        for c in ctxs:
            c.set_option("dist_timing", 1)
        k = min(steps, 16)
        per_rank = [[0.0, 0.0] for _ in plans]
        for i in range(k):
            engine.bfs_dist_run(plans, roots[i], -1, False)
            for r, p in enumerate(plans):
                a_, b_, c_ = p.dist_times()
                per_rank[r][0] += a_; per_rank[r][1] += b_
                if r == 0:
                    lm += a_; cm += b_; nl += c_
        for c in ctxs:
            c.set_option("dist_timing", 0)
        out = {"workload": f"RMAT scale-{scale} BFS, adjacency in {ndev} column slab(s) balanced by nnz, ONE process driving {ndev} "
                           f"GPUs, frontier words stored into the peers' bitmaps over xGMI (no RCCL call in the level loop)",
               "scale": scale, "vertices": int(n), "edges": int(nnz), "ranks": ndev, "TEPS": round(tot / dt, 1), "steps": steps,
               "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "scaling": "strong", "build_seconds": round(t_build, 2),
               "slab_nnz": slab_nnz, "levels_per_search": round(nl / max(k, 1), 2),
               "per_search_ms": {"level_kernels": round(lm / max(k, 1), 4), "frontier_exchange": round(cm / max(k, 1), 4)},
               "per_rank_ms_per_search": [{"rank": r, "level_kernels": round(x[0] / max(k, 1), 4),
                                           "frontier_exchange": round(x[1] / max(k, 1), 4)} for r, x in enumerate(per_rank)]}
        for p in plans:
            p.free()
        for m_ in keep:
            m_.free()
        return out
    finally:
        for c in ctxs:
            try:
                c.close()
            except Exception:
                pass


def bfs_gang_leg_guarded(engine, args, scale, ndev, torch, steps, warmup, deadline_s=240):
    """bfs_gang_leg behind a deadline and a try (never run on more than one real GPU: DESIGN.md §6) — it may cost the line
    this one secondary entry, never the headline."""
    import threading
    box = {}

    def work():
        try:
            box["d"] = bfs_gang_leg(engine, args, scale, ndev, torch, steps, warmup, args.opt)
        except BaseException as e:   # noqa: BLE001 — reported in the line
            box["err"] = repr(e)[:300]
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(deadline_s)
    if th.is_alive():
        STALLED_THREADS.append("peer-store BFS gang leg")
        return None, f"did not finish within {deadline_s} s"
    if "err" in box:
        return None, box["err"]
    return box["d"], None


def bfs_dist_leg_guarded(ctx, engine, args, scale, rank, world, dev, td, torch, steps, warmup, deadline_s=300):
    """bfs_dist_leg behind a deadline and a try: the RCCL exchange of this leg has never run on more than one real GPU
    (DESIGN.md §6), and a collective that stalls — or a rank that fails while the others wait for it — must cost the line this
    one secondary entry, not the headline.  Returns (result | None, error | None); after a stall the caller leaves through
    os._exit once its line is out (the worker thread sits inside the library and cannot be cancelled)."""
    import threading
    box = {}

    def work():
        try:
            if dev is not None and getattr(dev, "type", "") == "cuda":
                torch.cuda.set_device(dev)               # torch's current device is per thread
            box["d"] = bfs_dist_leg(ctx, engine, args, scale, rank, world, dev, td, torch, steps, warmup)
        except BaseException as e:   # noqa: BLE001 — reported in the line
            box["err"] = repr(e)[:300]
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(deadline_s)
    if th.is_alive():
        STALLED_THREADS.append("RCCL BFS leg")
        return None, f"did not finish within {deadline_s} s"
    if "err" in box:
        DIST_FAILED.append(box["err"])
        return None, box["err"]
    return box["d"], None


def emit(line, detail):
    """DETAIL line + sidecar first, the compact bench line LAST (the driver keeps the tail of stdout)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)        # RCCL's banner goes through C stdio: flush it before the JSON lines
    except Exception:
        pass
    blob = json.dumps(detail)
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(blob)
                break
        except OSError:
            continue
    print("DETAIL " + blob, flush=True)
    def slim(x):                                  # rates as integers: 7360579609797.9 TEPS says nothing 7360579609798 does not
        if isinstance(x, dict):
            return {k: slim(v) for k, v in x.items()}
        if isinstance(x, list):
            return [slim(v) for v in x]
        if isinstance(x, float) and abs(x) >= 1e6:
            return int(round(x))
        return x
    line = slim(line)
    txt = json.dumps(line, separators=(",", ":"))
    # the driver keeps the tail of stdout: the line must stay under 4 KB.  Secondary figures leave in this order until it does
    # (all of them are in the DETAIL line / bench_detail.json); what was dropped is named in the line.
    drop = [("khop22", "pinned_probe_ms"), ("khop22", "snapshot_prep_ms"), ("khop22", "prep_ms"), ("bfs22", "cpu_quartiles"),
            ("bfs22", "push_traffic"), ("khop24", "batch1024_dirty_ms"), ("khop26", "batch1024_dirty_ms"), ("khop26", "hop3_fetch_raw"),
            ("khop24", "cpu_TEPS"), ("khop26", "cpu_TEPS"), ("khop24", "batches"), ("khop26", "batches"), ("khop24", "parity_rows"),
            ("materialised24", "three_hop_64_rows"), ("bfs22", "host_arrays_ms"), ("bfs22", "push_frac"), ("bfs22", "pull_frac"),
            ("spmv_full_pass", "24"), ("spmv_full_pass", "26"), ("config5_varlen_TEPS",), ("operator20",), ("khop22", "emit3hop_entries"),
            ("bfs26",), ("materialised24",), ("khop24",)]
    dropped = []
    sec = line.get("secondary") if isinstance(line.get("secondary"), dict) else None
    while len(txt) >= 4000 and sec is not None and drop:
        path = drop.pop(0)
        node = sec
        for k in path[:-1]:
            node = node.get(k) if isinstance(node, dict) else None
        if isinstance(node, dict) and path[-1] in node:
            del node[path[-1]]
            dropped.append(".".join(path))
            line["secondary_dropped_for_length"] = dropped
            txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) < 4096, f"bench line is {len(txt)} bytes; the driver keeps only the tail of stdout"
    print(txt, flush=True)
    if STALLED_THREADS:                       # threads still spinning inside the library: do not join them at interpreter exit
        sys.stderr.write("bench.py: leaving through os._exit, stalled: %s\n" % ", ".join(STALLED_THREADS))
        sys.stderr.flush()
        os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16, help="timed whole-frontier label scans of the k-hop MATCH (per rank)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed scans before them")
    ap.add_argument("--scale", type=int, default=0, help="R-MAT scale of the headline leg (default 22)")
    ap.add_argument("--sources-per-call", type=int, default=0,
                    help="cap on the sources of one whole-frontier call (0 = the whole label, ~N/16; tools use it at scales 24 / 26, "
                         "where a whole label is 3.6 / 57 s per step)")
    ap.add_argument("--leg", default="khop", choices=("khop", "bfs"),
                    help="khop (default): the bench line.  bfs: only the BFS leg at --scale, its TEPS as `value` (tools; with "
                         "--force-dist the multi-rank slab path on one rank)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--alpha", type=float, default=0.0, help="BFS push->pull switch factor (0 = library default)")
    ap.add_argument("--force-dir", type=int, default=0, help="BFS: 0 auto, 1 push only, 2 pull only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of the secondary CPU BFS baseline")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="--leg bfs: run the multi-rank code path even with one rank")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (fgpu_set_option)")
    ap.add_argument("--quick", action="store_true", help="headline + parity + CPU baseline only (no secondary legs, no PMC)")
    ap.add_argument("--khop-extra-scales", default="24,26", help="further scales of the k-hop leg (secondary)")
    ap.add_argument("--bfs-threads", type=int, default=0, help="--leg bfs: also run the searches from this many query threads")
    ap.add_argument("--no-lanes-sweep", action="store_true", help="skip the run of the timed batches from 2 / 3 / 4 query threads")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle checks")
    ap.add_argument("--no-bfs", action="store_true", help="skip the BFS / SpMV secondary legs")
    ap.add_argument("--no-varlen", action="store_true", help="skip the config-5 stand-in leg")
    ap.add_argument("--no-operator", action="store_true", help="skip the CondTraverseOp::expand_batch vs fgpu_expand leg (RMAT-20, ~1 min)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (traffic = committed / null)")
    ap.add_argument("--bfs-steps", type=int, default=64)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # smoke test of the N > 1 control flow on a ONE-GPU box (tests / tools only, never a measurement): every rank on device
    # 0, torch collectives over gloo with host tensors, the RCCL BFS leg skipped (RCCL refuses two ranks on one device)
    one_device = os.environ.get("FGPU_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            # `python bench.py --gpus N` from a plain shell: become the launcher — one rank per GPU under
            # torch.distributed.run on this node, same arguments, same stdout (rank 0 prints the ONE line)
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            sys.stdout.flush()
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                      "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                      "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]])
        args.gpus = world

    import torch

    from falkordb_amd import engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    td = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as td
        if one_device:
            td.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    rdev = torch.device("cpu") if one_device else dev       # where the reduction tensors live

    def fence():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=rdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=rdev)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return int(t.item())

    ctx = engine.Context(local_rank)
    info = ctx.device_info()
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    base = {"n_gpus": world, "higher_is_better": True, "vs_baseline": None, "dtype": "u32", "data": "synthetic"}

    # ================= --leg bfs: the BFS leg alone (tools; --force-dist = the slab path on one rank) =================
    if args.leg == "bfs":
        scale = args.scale or (26 if world > 1 else 22)
        if use_dist:
            d, derr = bfs_dist_leg_guarded(ctx, engine, args, scale, rank, world, dev, td, torch, args.steps, args.warmup)
            if d is None:
                if rank == 0:
                    emit(dict(base, metric="traversed edges/sec (TEPS) on BFS (boolean vxm frontier loop), synthetic R-MAT", value=None,
                              unit="TEPS", steps=args.steps, warmup=args.warmup, ms_per_step=None, error=derr), {"bfs": {"error": derr}})
                os._exit(1)
        else:
            d, A, At, roots, _ = bfs_single_leg(ctx, engine, args, scale, steps=args.steps, warmup=args.warmup)
        if rank == 0:
            line = dict(base, metric="traversed edges/sec (TEPS) on BFS (boolean vxm frontier loop), synthetic R-MAT",
                        value=d["TEPS"], unit="TEPS", steps=args.steps, warmup=args.warmup, ms_per_step=d["ms_per_step"],
                        scaling="strong" if world > 1 else "weak",
                        config={"workload": d["workload"], "scale": scale, "device": info["name"]},
                        roofline=None, cpu_baseline=None)
            if d.get("query_threads"):
                line["query_threads"] = d["query_threads"]
            r = d.get("roofline")
            if r:
                line["roofline"] = {"bound": "hbm", "kernel": r["kernel"], "achieved": r["achieved"], "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": r["frac"], "traffic": None,
                                    "alg_bytes_per_launch": r["alg_bytes_per_launch"], "avg_launch_us": r["avg_launch_us"]}
            emit(line, {"bfs": d})
        if use_dist:
            td.barrier()
            td.destroy_process_group()
        return

    # ================= the bench line: k-hop MATCH =====================================================================
    scale = args.scale or 22
    line, head, (A, dp, dm, host, batch0, extra) = khop_headline(ctx, engine, args, scale, rank, world, fence, reduce_max,
                                                                 reduce_sum)
    detail = {"headline": head, "device": info}
    sec = {}
    roofline = parity = cpu = None
    if rank == 0:
        first, roofline, scan0 = extra
        key = "khop%d" % scale
        sec[key] = {"count_only_TEPS": head["count_only"]["TEPS"], "count_only_ms": head["count_only"]["ms_per_call"],
                    "dirty_TEPS": head["dirty"]["TEPS"], "dirty_ms": head["dirty"]["ms_per_call"],
                    "batch1024_TEPS": head["batch_1024"]["TEPS"], "batch1024_ms": head["batch_1024"]["ms_per_batch"]}
        if head.get("prep_ms") is not None:
            sec[key]["prep_ms"] = head["prep_ms"]     # first call of the process: pools + transpose, item lists, layout
            sec[key]["snapshot_prep_ms"] = head.get("snapshot_prep_ms")   # a new matrix version, pools warm
        if (head.get("pinned_probe") or {}).get("ms_per_batch"):
            sec[key]["pinned_probe_ms"] = head["pinned_probe"]["ms_per_batch"]
        em3 = head.get("emitting_three_hop_1024_rows") or {}
        if em3.get("ms_device"):
            sec[key].update({"emit3hop_ms_device": em3["ms_device"], "emit3hop_ms_host": em3["ms_host_arrays"], "emit3hop_entries": em3["entries"]})
        if head.get("lanes_sweep"):                   # the same 32 K sources in one call on 1 .. 4 lanes
            sec[key]["lanes_TEPS"] = {str(q["lanes"]): q["TEPS"] for q in head["lanes_sweep"]}
        if not args.no_parity and host is not None:
            parity, cpu = khop_parity_and_cpu(ctx, engine, args, A, dp, dm, host, batch0, first, scale)
            # the WHOLE :P scan — timed step 0 — against the committed oracle run of exactly these inputs
            # (tests/golden/make_khop22_scan_golden.py: a quarter of an hour of 16-thread CPU, done once)
            gpath = os.path.join(ROOT, "tests", "golden", "khop%d_scan.json" % scale)
            if os.path.exists(gpath) and args.edge_factor == 16 and scan0 is not None and args.sources_per_call <= 0:
                gold = json.load(open(gpath))
                ok = (gold.get("edges") == head["edges"] and gold.get("rows") == head["label_P_sources"] and
                      tuple(scan0) == (gold["nnz"], gold["checksum"], gold["flops"]))
                parity["whole_scan_vs_committed_oracle_run"] = {"ok": bool(ok), "rows": int(gold.get("rows", 0)),
                                                                "golden": "tests/golden/khop%d_scan.json" % scale}
                if not ok:
                    raise SystemExit(f"bench.py: whole-scan parity FAILED at scale {scale} against {gpath}: gpu {scan0} vs "
                                     f"({gold['nnz']}, {gold['checksum']}, {gold['flops']})")
                parity["ok"] = bool(parity["ok"] and ok)
                parity["rows"] = int(gold["rows"])
            detail["parity"] = parity
            if args.no_cpu_baseline:
                cpu = None
    host = None
    full = rank == 0 and world == 1 and not args.quick
    pmc = None
    bfs22 = None
    if full:
        # ---- BFS + boolean SpMV at the headline scale (BASELINE config 2 when it is 22), on the same adjacency -------
        if not args.no_bfs:
            bfs22, _, At, roots, _ = bfs_single_leg(ctx, engine, args, scale, A=A, steps=args.bfs_steps, warmup=8)
            if not args.no_cpu_baseline:
                bfs22["cpu_baseline"] = cpu_bfs_baseline(args, A, At, roots, scale, args.cpu_seconds)
            At.free()
            detail["bfs%d" % scale] = bfs22
    dp.free(); dm.free(); A.free()
    if full:
        # ---- k-hop at the other scales (BASELINE config 3 = RMAT-24; the metric's RMAT-26) ---------------------------
        for sc in [int(x) for x in args.khop_extra_scales.split(",") if x.strip() and int(x) != scale]:
            leg, (KA, Kdp, Kdm, Khost, kb) = khop_leg(ctx, engine, args, sc, 16 if sc <= 24 else 4,
                                                      parity_rows=1024 if sc <= 24 else 128, scan_sources=16384 if sc <= 24 else 8192)
            if sc == 24:
                if not args.no_roofline:
                    KAt = KA.transpose()
                    leg["spmv_full_pass"] = spmv_pass(ctx, engine, KAt, sc, iters=20)
                    KAt.free()
                detail["khop_materialised"] = khop_emit_leg(ctx, engine, args, KA, Khost, kb)
            Kdp.free(); Kdm.free(); KA.free()
            del Khost
            detail["khop%d" % sc] = leg
            r_ = leg.get("roofline") or {}
            scn = leg.get("scan") or {}
            sec["khop%d" % sc] = {"TEPS": (scn.get("clean") or leg["clean"])["TEPS"], "sources_per_call": scn.get("sources"),
                                  "ms_per_1024_sources": (scn.get("clean") or {}).get("ms_per_1024_sources"),
                                  "dirty_TEPS": (scn.get("dirty") or leg["dirty"])["TEPS"],
                                  "batch1024_TEPS": leg["clean"]["TEPS"], "batch1024_ms": leg["clean"]["ms_per_batch"],
                                  "batch1024_dirty_ms": leg["dirty"]["ms_per_batch"],
                                  "batches": leg["batches_timed"], "hop3_frac": r_.get("frac"), "hop3_us": r_.get("avg_launch_us"), "hop3_hop_frac": r_.get("hop_frac"),
                                  "parity_ok": leg["parity"].get("ok"), "parity_rows": leg["parity"].get("rows"),
                                  "cpu_TEPS": (leg.get("cpu_baseline") or {}).get("value")}
        # ---- RMAT-26 BFS on this one GPU: base point of the multi-GPU curve (BASELINE config 4) ----------------------
        if not args.no_bfs and scale != 26:
            b26, A26, At26, _, _ = bfs_single_leg(ctx, engine, args, 26, steps=32, warmup=8, want_prof=False)
            At26.free(); A26.free()
            detail["bfs26"] = b26
        # ---- BASELINE config 5's stand-in -----------------------------------------------------------------------------
        if not args.no_varlen:
            fl_v, dt_v, nb_v, un_v = varlen_leg(ctx, engine, args, 0, 1)
            detail["config5_varlen"] = {
                "workload": "BASELINE config 5 stand-in (LDBC SF100 is not available offline): R-MAT scale 19, edge factor 38 "
                            "(0.5 M vertices, ~20 M edges), `[*1..4]` reach sets (fgpu_expand_levels) from 1024-source batches "
                            "of label :P, 0.1 % tombstones + pending adds on every hop",
                "batches": int(nb_v), "TEPS": round(fl_v / dt_v, 1), "ms_per_batch": round(dt_v / max(nb_v, 1) * 1e3, 3),
                "distinct_pairs": int(un_v)}
            sec["config5_varlen_TEPS"] = detail["config5_varlen"]["TEPS"]
        # ---- the operator beside the bare device call ----------------------------------------------
        if not args.no_operator:
            op = operator_child(20)
            detail["operator20"] = op
            sec["operator20"] = {k_: op[k_] for k_ in ("rows_out", "ms_operator", "ms_bare_fgpu_expand", "ratio") if k_ in op} or op
        # ---- HBM traffic per launch, measured now: rocprofv3 --pmc passes over a reduced replay -----------------------
        if not args.no_pmc and not args.no_roofline:
            pmc = live_pmc(args)
            detail["pmc"] = pmc

            def hbm(name):
                e = pmc.get(name) if isinstance(pmc, dict) else None
                return e["hbm_bytes_per_dispatch"] if e else None
            if roofline:
                roofline["traffic"] = hbm(roofline["kernel"])
                roofline["traffic_source"] = ("live rocprofv3 --pmc passes over `bench.py --pmc-child`: FETCH_SIZE x2 + WRITE_SIZE per launch"
                                              if "error" not in pmc else "live PMC passes failed: " + str(pmc["error"])[:120])
            # the metric's other scale: the same passes over the RMAT-26 k-hop replay (no BFS part), hop 3's bytes per launch
            if "khop26" in sec and "error" not in pmc:
                import types
                pmc26 = live_pmc(types.SimpleNamespace(scale=26, no_bfs=True), timeout_s=240)
                detail["pmc_khop26"] = pmc26
                e26 = (pmc26.get("xp_stream_kernel") or pmc26.get("bp_pull_kernel<dense, count>")) if isinstance(pmc26, dict) else None
                if e26:
                    sec["khop26"]["hop3_traffic"] = e26["hbm_bytes_per_dispatch"]
                    sec["khop26"]["hop3_fetch_raw"] = e26["fetch_bytes_raw"]
            if bfs22 and bfs22.get("roofline"):
                for d in bfs22["roofline"]["by_direction"]:
                    d["traffic"] = hbm(d["kernel"])
            if bfs22 and bfs22.get("spmv_full_pass"):
                bfs22["spmv_full_pass"]["traffic"] = hbm(bfs22["spmv_full_pass"]["kernel"])
    elif world > 1:
        # ---- N > 1: the k-hop metric at its OTHER scale, RMAT-26 (north_star names it), the same way: every rank scans its own
        # slice of a label over a replicated adjacency (4.3 GB), no collective; the sums meet in one all-reduce afterwards -------
        if not args.quick and not one_device:
            try:
                A26 = ctx.mat_rmat(26, args.edge_factor, 0x5EED1234 + 26)
                S26 = 8192
                lab = p_label_sources(A26.nrows, rank % 16)[:S26]
                engine.expand_count(ctx, lab[:2048], [A26] * 3)
                engine.expand_count(ctx, lab, [A26] * 3)
                fence()
                t_ = time.perf_counter()
                r26 = engine.expand_count(ctx, lab, [A26] * 3)
                fence()
                d26 = reduce_max(time.perf_counter() - t_)
                f26 = reduce_sum(int(r26[2]))
                A26.free()
                sec["khop26_weak"] = {"TEPS": round(f26 / d26, 1), "ms": round(d26 * 1e3, 3), "sources_per_rank": S26, "ranks": world,
                                      "scaling": "weak"}
            except Exception as e:   # noqa: BLE001 — a secondary figure
                sec["khop26_weak"] = {"error": repr(e)[:160]}
                fence()
        # ---- BASELINE config 4 as a secondary leg of the same line -----------------------------------------------------
        if not args.no_bfs and not one_device:
            d, derr = bfs_dist_leg_guarded(ctx, engine, args, 26, rank, world, dev, td, torch, 32, 8)
            if d is None:
                detail["bfs26_dist"] = {"error": derr}
                sec["bfs26_dist"] = {"error": derr[:160], "ranks": world}
            else:
                detail["bfs26_dist"] = d
                sec["bfs26_dist"] = {"TEPS": d["TEPS"], "ms": d["ms_per_step"], "ranks": world, "rccl_ranks": ctx.comm_info()[1],
                                     "scaling": "strong", "levels": d["levels_per_search"],
                                     "level_kernels_ms": d["per_search_ms"]["level_kernels"],
                                     "frontier_exchange_ms": d["per_search_ms"]["frontier_exchange"]}
            # the same searches with ONE process driving all the GPUs and peer stores instead of RCCL calls (rank 0 only; the other
            # ranks wait on the launcher's key-value store, not in a device-side barrier that would hold CUs of their GPUs)
            store = None
            try:
                store = td.distributed_c10d._get_default_store()
            except Exception:
                pass
            if rank == 0:
                g, gerr = (None, "skipped: the RCCL leg left a stalled thread") if STALLED_THREADS else \
                    bfs_gang_leg_guarded(engine, args, 26, world, torch, 32, 8)
                if g is None:
                    detail["bfs26_gang"] = {"error": gerr}
                    sec["bfs26_gang"] = {"error": gerr[:160], "ranks": world}
                else:
                    detail["bfs26_gang"] = g
                    sec["bfs26_gang"] = {"TEPS": g["TEPS"], "ms": g["ms_per_step"], "ranks": world, "scaling": "strong",
                                         "exchange": "peer stores, one process", "levels": g["levels_per_search"],
                                         "level_kernels_ms": g["per_search_ms"]["level_kernels"],
                                         "frontier_exchange_ms": g["per_search_ms"]["frontier_exchange"]}
                # the 1-GPU base point of BOTH BFS-26 legs, on rank 0's device, in the same line: the >= 6 x of north_star can be read
                # off one SCALE record (bfs26_gang.TEPS / bfs26_base.TEPS, bfs26_dist.TEPS / bfs26_base.TEPS)
                if not STALLED_THREADS:
                    try:
                        nr = types_ns(no_roofline=True, edge_factor=args.edge_factor, alpha=args.alpha, force_dir=args.force_dir)
                        b_ = strong_scaling_base(ctx, engine, nr, 26, steps=16, warmup=4)
                        detail["bfs26_base"] = b_
                        sec["bfs26_base"] = {"TEPS": b_["value"], "ms": b_["ms_per_step"], "ranks": 1}
                        for k_ in ("bfs26_dist", "bfs26_gang"):
                            if isinstance(sec.get(k_), dict) and sec[k_].get("TEPS"):
                                sec[k_]["x_base"] = round(sec[k_]["TEPS"] / max(b_["value"], 1.0), 3)
                    except Exception as e:   # noqa: BLE001
                        sec["bfs26_base"] = {"error": repr(e)[:160]}
                if store is not None:
                    try:
                        store.set("fgpu_gang_done", "1")
                    except Exception:
                        pass
            elif store is not None:
                try:
                    import datetime
                    store.wait(["fgpu_gang_done"], datetime.timedelta(seconds=600))
                except Exception:
                    pass
    if rank == 0:
        if bfs22:
            r = bfs22.get("roofline") or {}
            dirs = {("push" if "push" in d["kernel"] else "pull"): d for d in r.get("by_direction", [])}
            sec["bfs%d" % scale] = {"TEPS": bfs22["TEPS"], "ms": bfs22["ms_per_step"],
                                    "push_frac": (dirs.get("push") or {}).get("frac"), "pull_frac": (dirs.get("pull") or {}).get("frac"),
                                    "push_traffic": (dirs.get("push") or {}).get("traffic"),
                                    "host_arrays_ms": bfs22["host_arrays"]["ms_per_step"],
                                    "cpu_TEPS": (bfs22.get("cpu_baseline") or {}).get("value"),
                                    "cpu_quartiles": (bfs22.get("cpu_baseline") or {}).get("quartiles")}
            if not args.no_lanes_sweep:
                q2 = bfs_threads_child(args, scale)
                detail["bfs%d_query_threads" % scale] = q2
                if q2.get("TEPS"):
                    sec["bfs%d" % scale].update({"threads%d_TEPS" % q2["threads"]: q2["TEPS"], "threads%d_ms" % q2["threads"]: q2["ms_per_step"]})
        if detail.get("bfs26"):
            sec["bfs26"] = {"TEPS": detail["bfs26"]["TEPS"], "ms": detail["bfs26"]["ms_per_step"],
                            "host_arrays_ms": detail["bfs26"]["host_arrays"]["ms_per_step"]}
        spm = {}
        for key in ("bfs%d" % scale, "khop24", "bfs26"):
            sp = (detail.get(key) or {}).get("spmv_full_pass")
            if sp:
                spm[str(sp["scale"])] = {"kernel": sp["kernel"].split("_")[0], "frac_cold": sp["frac"], "frac_warm": sp["warm"]["frac"],
                                         "us_cold": sp["avg_launch_us"], "traffic": sp.get("traffic")}
        if spm:
            sec["spmv_full_pass"] = spm
        em = detail.get("khop_materialised")
        if em:
            sec["materialised24"] = {k: {"ms_device": v["ms_per_batch"], "ms_host_arrays": v["host_arrays_ms"],
                                         "ms_host_arrays_u64": v.get("host_arrays_u64_ms"),
                                         "ms_stream_all": v["stream_ms"]["all_chunks"],
                                         "entries": v["out_nnz_per_batch"], "parity_ok": (v.get("parity") or {}).get("ok")}
                                     for k, v in em.items()}
        out = dict(base, metric=BASELINE_METRIC, value=line["value"], unit="TEPS", steps=args.steps, warmup=args.warmup,
                   ms_per_step=line["ms_per_step"], scaling="weak",
                   config={"workload": f"RMAT scale-{scale} 3-hop MATCH (a:P)-->()-->()-->(c): one whole-frontier fgpu_expand_count call "
                                       f"per label scan, clean layers, count + checksum",
                           "scale": scale, "vertices": head["vertices"], "edges": head["edges"], "hops": 3,
                           "sources_per_call": head["scan"]["sources_per_call"], "live_sources": head["scan"]["live_sources_last_call"],
                           "pass_rows": head["scan"]["pass_rows"], "passes_per_call": head["scan"]["passes_last_call"],
                           "lanes": head["scan"]["lanes"],
                           "edge_factor": args.edge_factor, "device": info["name"],
                           "parallelism": ("1 GPU" if world == 1 else f"{world} GPUs: label scans round-robin over the ranks, "
                                                                      f"adjacency replicated, no collective")},
                   roofline=roofline, cpu_baseline=cpu,
                   parity=({"ok": parity["ok"], "rows": parity["rows"],
                            "what": "(nnz, checksum, flops): first 1024 :P rows clean + dirty vs the oracle run here"
                                    + ("; timed step 0 (whole :P scan) vs tests/golden/khop%d_scan.json" % scale
                                       if parity.get("whole_scan_vs_committed_oracle_run") else "")} if parity else {"checked": False}),
                   secondary=sec)
        emit(out, detail)
    if use_dist:
        if STALLED_THREADS or DIST_FAILED:     # other ranks may never reach the barrier (or this one left a thread behind)
            sys.stdout.flush()
            os._exit(0)
        try:
            td.barrier()
            td.destroy_process_group()
        except Exception as e:   # noqa: BLE001 — a peer that left early (its own stall or failure) must not turn this rank's exit code
            sys.stderr.write("bench.py: rank %d: closing barrier failed (%s); leaving\n" % (rank, repr(e)[:200]))
            sys.stderr.flush()
            sys.stdout.flush()
            os._exit(0)


BASELINE_METRIC = "traversed edges/sec (TEPS) on k-hop MATCH, RMAT scale-22/26; % HBM roofline"


if __name__ == "__main__":
    main()
