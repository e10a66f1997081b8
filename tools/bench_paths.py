"""Secondary hot-path measurements (one JSON line each; bench.py stays the BFS headline):

  merge   Delta merge (m \\ dm) U dp on RMAT-<scale> with 0.1 % pending adds + 0.1 % tombstones
          (SURVEY §8d config 5 "merge GB/s"; K6): entry-parallel kernel vs the wavefront-per-row one,
          pattern and UINT64 layers
  expand  k-hop CondTraverse core (fgpu_expand_count) on RMAT-<scale>, 1024-row batches, clean and dirty
          layers
  host    CondTraverseOp::expand_batch through the C++ host layer vs the bare fgpu_expand call

  reach   [*1..4] DISTINCT reachability (fgpu_expand_levels) with dirty layers, device fold, folded rerun
          (SURVEY §8d config 5 stand-in)

  pagerank  algo.pageRank's core (fgpu_pagerank, FP32 plus_second pull SpMV) on RMAT-<scale>: ms per iteration

usage: python tools/bench_paths.py [merge|expand|reach|host|pagerank|all] [scale]
"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from falkordb_amd import engine


def timed(ctx, fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def deltas(ctx, A, frac, rng, valued=False):
    n, nnz = A.nrows, A.nvals
    k = max(1, int(nnz * frac))
    # tombstones: a uniformly random `frac` of the stored entries, drawn on the device (fgpu_mat_sample) — the first
    # version took them from rows [0, 65536] only, which clustered the touched rows and flattered the clean-word
    # merge path; pending adds: k uniformly random coordinates
    dm = A.sample(int(rng.integers(1, 1 << 30)), max(1, int(round(1.0 / frac))))
    pr = rng.integers(0, n, k, dtype=np.uint64)
    pc = rng.integers(0, n, k, dtype=np.uint64)
    dp = ctx.mat_from_coo(n, n, pr, pc, rng.integers(0, 1 << 40, k, dtype=np.uint64) if valued else None)
    return dp, dm


def bench_merge(ctx, scale):
    rng = np.random.default_rng(1)
    A = ctx.mat_rmat(scale)
    n, nnz = A.nrows, A.nvals
    dp, dm = deltas(ctx, A, 0.001, rng)
    for mode in (0, 2, 1):
        ctx.set_option("merge_mode", mode)
        dt, out = timed(ctx, lambda: A.merge(dp, dm))
        b_alg = 4 * (nnz + dp.nvals + dm.nvals) + 4 * out.nvals + 8 * (n + 1)
        print(json.dumps({"path": "delta_merge", "layers": "bool",
                          "kernel": ["entry-parallel", "row-wave", "entry-parallel, base marked per entry"][mode],
                          "scale": scale, "nnz_m": nnz, "nnz_dp": dp.nvals, "nnz_dm": dm.nvals,
                          "nnz_out": out.nvals, "ms": round(dt * 1e3, 3), "alg_bytes": b_alg,
                          "GBps": round(b_alg / dt / 1e9, 1), "frac_hbm": round(b_alg / dt / 8e12, 4)}), flush=True)
    ctx.set_option("merge_mode", 0)
    # UINT64 layers (Tensor::flush): values ride along, 8 B per entry each way
    rp, ci, _ = A.export_csr()
    r = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp).astype(np.int64))
    V = ctx.mat_from_coo(n, n, r, ci, np.arange(len(ci), dtype=np.uint64))
    del r, ci, rp
    dpv, dmv = deltas(ctx, A, 0.001, rng, valued=True)
    dt, out = timed(ctx, lambda: V.merge(dpv, dmv), reps=3)
    b_alg = 12 * (nnz + dpv.nvals) + 4 * dmv.nvals + 12 * out.nvals + 8 * (n + 1)
    print(json.dumps({"path": "delta_merge", "layers": "u64", "kernel": "entry-parallel", "scale": scale,
                      "nnz_m": nnz, "nnz_out": out.nvals, "ms": round(dt * 1e3, 3), "alg_bytes": b_alg,
                      "GBps": round(b_alg / dt / 1e9, 1), "frac_hbm": round(b_alg / dt / 8e12, 4)}), flush=True)
    dt, out = timed(ctx, lambda: V.merge_pattern(dpv, dmv), reps=3)
    print(json.dumps({"path": "tensor_extract", "kernel": "entry-parallel", "scale": scale, "nnz_out": out.nvals,
                      "ms": round(dt * 1e3, 3)}), flush=True)
    dt, out = timed(ctx, lambda: A.transpose(), reps=3)
    print(json.dumps({"path": "transpose", "layers": "bool", "scale": scale, "ms": round(dt * 1e3, 3)}), flush=True)


def bench_expand(ctx, scale, hops=3, batch=1024, nbatches=3, rank=0, nranks=1, device="cpu"):
    """Under torch.distributed.run (WORLD_SIZE > 1) the batch is `batch` rows PER RANK, sharded by
    dist.expand_count_sharded over replicated layers (weak scaling, no data-path collective)."""
    from falkordb_amd import dist as fdist
    rng = np.random.default_rng(7)
    A = ctx.mat_rmat(scale)
    n, nnz = A.nrows, A.nvals
    dp, dm = deltas(ctx, A, 0.001, rng)
    for name, layers, cs in (("clean", ([A] * hops, None, None), True),
                             ("clean", ([A] * hops, None, None), False),
                             ("dirty-0.1%", ([A] * hops, [dp] * hops, [dm] * hops), True),
                             ("dirty-0.1%", ([A] * hops, [dp] * hops, [dm] * hops), False)):
        tot_t, tot_f, tot_n = 0.0, 0, 0
        for b in range(nbatches + 1):
            src = rng.choice(n, batch * nranks, replace=False).astype(np.uint64)
            ctx.sync()
            if nranks > 1:
                import torch.distributed as td
                td.barrier()
            t0 = time.perf_counter()
            out_nnz, flops = fdist.expand_count_sharded(
                lambda s_: engine.expand_count(ctx, s_, *layers, want_checksum=cs)[::2], src, rank, nranks, device)
            dt = time.perf_counter() - t0
            if b == 0:
                continue  # warm-up (transpose cache, pools)
            tot_t += dt; tot_f += flops; tot_n += out_nnz
        if rank:
            continue
        print(json.dumps({"path": "khop_expand", "layers": name, "result": "count + checksum" if cs else "count only",
                          "n_gpus": nranks, "scale": scale, "hops": hops, "batch_rows": batch * nranks,
                          "batches": nbatches, "ms_per_batch": round(tot_t / nbatches * 1e3, 3),
                          "flops_per_batch": tot_f // nbatches, "out_nnz_per_batch": tot_n // nbatches,
                          "GTEPS": round(tot_f / tot_t / 1e9, 2),
                          "alg_bytes_per_batch": int((4 * tot_f + 4 * tot_n) // nbatches),
                          "GBps": round((4 * tot_f + 4 * tot_n) / tot_t / 1e9, 1)}), flush=True)


def bench_reach(ctx, scale=19, edge_factor=38, hops=4, batch=1024):
    """BASELINE config 5 stand-in (LDBC SF100 is not available offline): ~0.5 M vertices / ~20 M edges,
    [*1..4] DISTINCT reachability of 1024 sources with dirty layers, the Delta fold on device, then the same
    query on the folded base."""
    rng = np.random.default_rng(5)
    A = ctx.mat_rmat(scale, edge_factor)
    n, nnz = A.nrows, A.nvals
    dp, dm = deltas(ctx, A, 0.001, rng)
    src = rng.choice(n, batch, replace=False).astype(np.uint64)
    for name, m, p_, d_ in (("dirty-0.1%", A, dp, dm), ("folded", None, None, None)):
        if m is None:
            dt_fold, m = timed(ctx, lambda: A.merge(dp, dm), reps=3)
            b_alg = 4 * (nnz + dp.nvals + dm.nvals) + 4 * m.nvals + 8 * (n + 1)
            print(json.dumps({"path": "delta_fold", "scale": scale, "nnz_m": nnz, "nnz_out": m.nvals,
                              "ms": round(dt_fold * 1e3, 3), "GBps": round(b_alg / dt_fold / 1e9, 1)}), flush=True)
        args = ([m] * hops, [p_] * hops if p_ else None, [d_] * hops if d_ else None)
        engine.expand_levels(ctx, src, *args)   # warm-up: transpose cache
        ctx.sync()
        t0 = time.perf_counter()
        r = engine.expand_levels(ctx, src, *args)
        dt = time.perf_counter() - t0
        print(json.dumps({"path": "varlen_reach", "layers": name, "scale": scale, "edges": m.nvals, "hops": hops,
                          "batch_rows": batch, "ms": round(dt * 1e3, 3), "hop_nnz": r["hop_nnz"],
                          "distinct_1_to_k": r["union_nnz"], "flops": r["flops"],
                          "GTEPS": round(r["flops"] / dt / 1e9, 2)}), flush=True)


def bench_pagerank(ctx, scale):
    """LAGr_PageRank's iteration on device: fixed 20 iterations (tol 0) so that the figure is per iteration;
    B_alg per iteration = 4 nnz (column ids of A') + 8 (N+1) (row pointers) + 6 x 4 N (t, d, w read / w, r written,
    sink bytes) with the gathers of w counted as cache traffic like the bitmap probes of the BFS formulas."""
    A = ctx.mat_rmat(scale)
    At = A.transpose()
    n, nnz = A.nrows, A.nvals
    engine.pagerank(ctx, A, At, None, 0.85, 0.0, 3)
    ctx.sync()
    its = 20
    t0 = time.perf_counter()
    scores, it = engine.pagerank(ctx, A, At, None, 0.85, 0.0, its)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    engine.pagerank(ctx, A, At, None, 0.85, 0.0, 3 * its)
    dt3 = time.perf_counter() - t0
    marginal = (dt3 - dt) / (2 * its)          # what one more iteration costs (the call's fixed part — set-up, the 4 N-byte copy-out — drops out)
    t1 = time.perf_counter()
    s2, it2 = engine.pagerank(ctx, A, At)
    dt2 = time.perf_counter() - t1
    b_alg = 4 * nnz + 8 * (n + 1) + 24 * n
    print(json.dumps({"path": "pagerank", "scale": scale, "vertices": n, "edges": nnz, "iterations": it,
                      "ms_per_iteration": round(dt / it * 1e3, 3), "ms_per_additional_iteration": round(marginal * 1e3, 3),
                      "alg_bytes_per_iteration": b_alg,
                      "GBps": round(b_alg * it / dt / 1e9, 1), "frac_hbm": round(b_alg * it / dt / 8e12, 4),
                      "default_run": {"tol": 1e-4, "iterations": it2, "ms": round(dt2 * 1e3, 3)},
                      "sum": round(float(scores.astype(np.float64).sum()), 6),
                      "note": "whole fgpu_pagerank call incl. one host sync per four iterations and the D2H of the scores"}), flush=True)


def bench_host(scale):
    """expand_batch through libfalkor_host.so (label probes, layer waits, result hand-off) vs bare fgpu_expand."""
    from falkordb_amd import host
    hc = host.Context(0)
    ctx = engine.Context(0)
    A = ctx.mat_rmat(scale)
    n = A.nrows
    rp, ci, _ = A.export_csr()
    r = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp).astype(np.int64))
    g = host.Graph(hc, n)
    t = g.add_type("KNOWS")
    lp = g.add_label("P")
    t0 = time.perf_counter()
    g.create_edges(t, r, ci, np.arange(len(ci), dtype=np.uint64))
    g.commit()
    build = time.perf_counter() - t0
    rng = np.random.default_rng(3)
    src = rng.choice(n, 1024, replace=False).astype(np.uint64)
    spec = host.cond_spec(hops=[(["KNOWS"], []), (["KNOWS"], [])])
    srcl = src.tolist()
    for _ in range(2):
        (rows, _, _), nulls, flops = g.cond_traverse_batch(spec, srcl, as_arrays=True)
    t0 = time.perf_counter()
    (rows, _, _), nulls, flops = g.cond_traverse_batch(spec, srcl, as_arrays=True)
    t_host = time.perf_counter() - t0
    import ctypes
    g.L.fh_last_op_ns.restype = ctypes.c_uint64
    t_cpp = g.L.fh_last_op_ns() / 1e9
    for _ in range(2):
        rowptr, dest, fl = engine.expand(ctx, src, [A, A])
    t0 = time.perf_counter()
    rowptr, dest, fl = engine.expand(ctx, src, [A, A])
    t_raw = time.perf_counter() - t0
    assert len(rows) == len(dest) and fl == flops
    print(json.dumps({"path": "host_expand_batch", "scale": scale, "hops": 2, "batch_rows": 1024, "rows_out": len(rows),
                      "flops": flops, "ms_cpp_expand_batch": round(t_cpp * 1e3, 3), "ms_through_ctypes": round(t_host * 1e3, 3), "ms_bare_fgpu_expand": round(t_raw * 1e3, 3),
                      "graph_load_s": round(build, 2),
                      "note": "cpp = CondTraverseOp::expand_batch alone (label probes, fgpu_expand, result columns); through_ctypes adds the test harness' copies into numpy"}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if what in ("merge", "all"):
        c = engine.Context(0)
        bench_merge(c, scale or 22)
        c.close()
    if what in ("expand", "all"):
        import os
        world, rank, dev = int(os.environ.get("WORLD_SIZE", "1")), 0, "cpu"
        if world > 1:    # python -m torch.distributed.run --nproc-per-node N tools/bench_paths.py expand 24
            import torch
            import torch.distributed as td
            rank = int(os.environ["RANK"])
            local = int(os.environ.get("LOCAL_RANK", rank))
            torch.cuda.set_device(local)
            dev = torch.device("cuda", local)
            td.init_process_group("nccl", device_id=dev)
            c = engine.Context(local)
        else:
            c = engine.Context(0)
        bench_expand(c, scale or 24, rank=rank, nranks=world, device=dev)
        c.close()
        if world > 1:
            td.destroy_process_group()
        if what == "expand":
            sys.exit(0)
    if what in ("reach", "all"):
        c = engine.Context(0)
        bench_reach(c, scale or 19)
        c.close()
    if what in ("pagerank", "all"):
        c = engine.Context(0)
        bench_pagerank(c, scale or 22)
        c.close()
    if what in ("host", "all"):
        bench_host(scale or 18)
