"""GPU: the multi-rank (column-slab) BFS kernels, exercised on ONE device by running `nranks` slab
plans side by side and doing the frontier all-gather by hand (device copies into each plan's
gather buffer).  The control flow is falkordb_amd.dist.run_levels — the same loop bench.py runs
over RCCL — so this pins the partitioned step / commit kernels against the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from falkordb_amd import dist as fdist
from falkordb_amd import engine

pytestmark = pytest.mark.gpu


class Gang:
    """`nranks` HipSlabBackends on one GPU stepping in lock-step (a stand-in for one rank per GPU)."""

    def __init__(self, ctx, A, nranks, dev, protocol="fused"):
        self.backs = []
        self.protocol = protocol
        n = A.nrows
        for r in range(nranks):
            lo, hi, _ = fdist.slab_range(n, r, nranks)
            a_slab = A.col_slab(lo, min(hi, n))
            self.backs.append(fdist.HipSlabBackend(ctx, a_slab, a_slab.transpose(), r, nranks, dev, protocol))
        self.nranks = nranks
        if protocol == "fused":
            # what torch.distributed.all_reduce does across processes: slab-local row degrees -> global
            total = self.backs[0].deg.clone()
            for b in self.backs[1:]:
                total += b.deg
            for b in self.backs:
                b.set_degrees(total)

    def begin(self, src, max_level=-1):
        for b in self.backs:
            b.begin(src, max_level)

    def step(self):
        for b in self.backs:
            b.step()

    def gather(self):
        if self.nranks == 1 and self.protocol == "stepped":
            return  # single-rank stepped plans work in place: local IS the global buffer
        wpr = self.backs[0].words_per_rank
        for dst in self.backs:
            for r, src in enumerate(self.backs):
                dst.glob[r * wpr:(r + 1) * wpr].copy_(src.send_buffer())

    def commit(self):
        for b in self.backs:
            b.commit()

    def done(self):
        res = [b.done() for b in self.backs]
        assert len(set(res)) == 1, f"ranks disagree on termination: {res}"
        return res[0]

    def levels(self, n):
        out = np.full(n, -1, dtype=np.int32)
        for r, b in enumerate(self.backs):
            lv, _ = b.plan.fetch()  # only the rank's own slab [lo, hi) is filled in
            lo, hi, _ = fdist.slab_range(n, r, self.nranks)
            out[lo:min(hi, n)] = lv[lo:min(hi, n)]
        return out


@pytest.mark.parametrize("protocol", ["fused", "stepped"])
@pytest.mark.parametrize("nranks", [1, 2, 4])
@pytest.mark.parametrize("force", [0, 1, 2])
def test_slab_partitioned_bfs_matches_oracle(ctx, nranks, force, protocol):
    scale = 14
    a = oracle.rmat_csr(scale)
    A = ctx.mat_rmat(scale)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        try:
            gang = Gang(ctx, A, nranks, dev, protocol)
            for b in gang.backs:
                b.plan.tune(force_direction=force)
            for src in [int(np.argmax(np.diff(a.rowptr))), 5]:
                for max_level in (-1, 2):
                    fdist.run_levels(gang, gang.gather, src, max_level)
                    ref, _, _ = oracle.bfs(a, src, max_level)
                    np.testing.assert_array_equal(gang.levels(a.nrows), ref)
                    if max_level < 0:
                        # slab-local out-degree sums add up to the global traversed-edge count
                        edges = sum(b.plan.stats()["edges_traversed"] for b in gang.backs)
                        assert edges == int(np.diff(a.rowptr)[ref >= 0].sum())
                        if protocol == "fused":   # the owner accounts a vertex: shares add up to the total
                            assert sum(b.plan.stats()["reached"] for b in gang.backs) == int((ref >= 0).sum())
        finally:
            torch.cuda.synchronize()
            ctx.set_stream(None)


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_row_sharded_expand_concatenates_to_the_whole_batch(ctx, nranks):
    """k-hop MATCH over ranks: shard the source rows with dist.shard_rows, run every share
    through fgpu_expand on replicated layers, concatenate in rank order == one fgpu_expand of the whole batch
    == the oracle's delta_lmxm chain."""
    a = oracle.rmat_csr(12)
    n = a.nrows
    rng = np.random.default_rng(3)
    rows, cols = a.pairs()
    pick = rng.choice(len(rows), 200, replace=False)
    dm = oracle.build_csr(n, n, rows[pick], cols[pick])
    dp = oracle.build_csr(n, n, rng.integers(0, n, 200).astype(np.uint64), rng.integers(0, n, 200).astype(np.uint64))
    up = lambda m: ctx.mat_from_csr(m.nrows, m.ncols, m.rowptr, m.colidx)
    A, DP, DM = up(a), up(dp), up(dm)
    src = rng.integers(0, n, 257).astype(np.uint64)
    local = lambda s: engine.expand(ctx, s, [A, A], [DP, DP], [DM, DM])
    parts = []
    for r in range(nranks):
        lo, hi = fdist.shard_rows(len(src), r, nranks)
        parts.append(fdist.expand_sharded(local, src[lo:hi], 0, 1))
    counts = np.concatenate([np.diff(p[0].astype(np.int64)) for p in parts])
    dest = np.concatenate([p[1] for p in parts])
    flops = sum(p[2] for p in parts)
    c = oracle.build_csr(len(src), n, np.arange(len(src), dtype=np.uint64), src)
    flops_ref = 0
    for _ in range(2):
        c, fl = oracle.delta_lmxm(c, a, dp, dm)
        flops_ref += fl
    np.testing.assert_array_equal(np.concatenate([[0], np.cumsum(counts)]), c.rowptr)
    np.testing.assert_array_equal(dest, c.colidx)
    assert flops == flops_ref


# ---- the in-library loop (fgpu_bfs_dist_run): level kernels + frontier exchange inside libfgpu.so ------------------

def _col_block_counts(a: oracle.CSR, shift):
    return np.bincount((a.colidx >> np.uint64(shift)).astype(np.int64), minlength=((a.ncols + (1 << shift) - 1) >> shift))


@pytest.mark.parametrize("scale,nparts", [(14, 1), (14, 2), (14, 3), (16, 8)])
def test_balanced_splits_follow_the_in_degree_prefix(ctx, scale, nparts):
    a = oracle.rmat_csr(scale)
    A = ctx.mat_rmat(scale)
    got = A.balanced_splits(nparts).tolist()
    shift = fdist.splits_shift(a.ncols)
    # the device histogram + the library's boundary choice against an independent restatement of the rule: boundary k is
    # the block edge whose entry prefix is nearest to k * nnz / nparts (ties go to the later edge), never moving back
    pre = np.concatenate([[0], np.cumsum(_col_block_counts(a, shift))])
    want, j = [0], 0
    for k in range(1, nparts):
        t = a.nnz * k / nparts
        while j < len(pre) - 1 and abs(pre[j + 1] - t) <= abs(pre[j] - t):
            j += 1
        want.append(min(j << shift, ((a.ncols + 4095) >> 12) << 12))
    want.append(((a.ncols + 4095) >> 12) << 12)
    assert got == want
    assert got == fdist.balanced_splits(_col_block_counts(a, shift), a.ncols, nparts, shift)   # the host half on its own
    assert got[0] == 0 and got[-1] == ((a.ncols + 4095) >> 12) << 12 and all(x % 4096 == 0 for x in got)
    assert all(x <= y for x, y in zip(got, got[1:]))
    # every part holds its share of the entries, to within one 4096-column block of in-edges
    cols = a.colidx.astype(np.int64)
    per = [int(((cols >= lo) & (cols < hi)).sum()) for lo, hi in zip(got, got[1:])]
    assert sum(per) == a.nnz
    biggest_block = int(_col_block_counts(a, shift).max())
    assert max(per) <= a.nnz / nparts + biggest_block


def _gang_plans(ctx, A, nranks, splits):
    plans, keep = [], []
    for r in range(nranks):
        a_slab = A.col_slab(int(splits[r]), int(min(splits[r + 1], A.ncols)))
        at_slab = a_slab.transpose()
        keep += [a_slab, at_slab]
        plans.append(engine.BfsPlan(ctx, a_slab, at_slab, r, nranks, splits=splits))
    return plans, keep


@pytest.mark.parametrize("nranks", [1, 2, 3, 4])
@pytest.mark.parametrize("force", [0, 1, 2])
def test_in_library_dist_loop_matches_the_oracle(ctx, nranks, force):
    """fgpu_bfs_dist_run with the plans of ALL ranks on one device (exchange = event-ordered copies): nnz-balanced,
    hence uneven, slabs; levels, parents and the per-rank shares of reached / edges_traversed against the oracle."""
    scale = 14
    a = oracle.rmat_csr(scale)
    A = ctx.mat_rmat(scale)
    n = a.nrows
    splits = A.balanced_splits(nranks)
    if nranks == 3:
        assert len({int(splits[r + 1] - splits[r]) for r in range(nranks)}) > 1   # 4 blocks over 3 ranks: uneven slabs
    plans, keep = _gang_plans(ctx, A, nranks, splits)
    for p in plans:
        p.tune(force_direction=force)
    deg = np.diff(a.rowptr)
    for src in [int(np.argmax(deg)), 5, int(np.nonzero(deg > 0)[0][-1])]:
        for max_level in (-1, 2):
            ctx.set_option("dist_timing", 1 if max_level < 0 else 0)   # the time split is opt-in (events cost stream time)
            engine.bfs_dist_run(plans, src, max_level, want_parent=True)
            ctx.set_option("dist_timing", 0)
            ref, _, ref_edges = oracle.bfs(a, src, max_level)
            level = np.full(n, -1, dtype=np.int32)
            parent = np.full(n, -1, dtype=np.int64)
            for r, p in enumerate(plans):
                lv, par = p.fetch(want_parent=True)
                lo, hi = int(splits[r]), int(min(splits[r + 1], n))
                level[lo:hi] = lv[lo:hi]
                parent[lo:hi] = par[lo:hi]
            np.testing.assert_array_equal(level, ref)
            reached = np.nonzero((ref > 0))[0]
            assert (ref[parent[reached]] + 1 == ref[reached]).all() and parent[src] == src
            s_ = a.to_set()
            assert all((int(parent[v]), int(v)) in s_ for v in reached[:500])
            if max_level < 0:
                st = [p.stats() for p in plans]
                assert sum(x["reached"] for x in st) == int((ref >= 0).sum())
                assert sum(x["edges_traversed"] for x in st) == ref_edges
            lm, cm, nl = plans[0].dist_times()
            assert nl >= int(ref.max()) and (lm > 0) == (max_level < 0)


def test_in_library_loop_over_an_rccl_communicator_of_one(ctx):
    """The RCCL side of the boundary on a 1-GPU box: ncclGetUniqueId / ncclCommInitRank inside libfgpu.so, a
    partitioned plan of one rank, fgpu_bfs_dist_run through the communicator."""
    ctx2 = engine.Context(0)
    try:
        uid = ctx2.comm_unique_id()
        assert len(uid) == 128
        ctx2.comm_init_rank(1, 0, uid)
        assert ctx2.comm_info() == (0, 1)
        a = oracle.rmat_csr(13)
        A = ctx2.mat_rmat(13)
        splits = A.balanced_splits(1)
        plan = engine.BfsPlan(ctx2, A, A.transpose(), 0, 1, splits=splits)
        for src in (3, int(np.argmax(np.diff(a.rowptr)))):
            engine.bfs_dist_run([plan], src)
            lv, _ = plan.fetch()
            np.testing.assert_array_equal(lv[:a.nrows], oracle.bfs(a, src, -1)[0])
        plan.free()
        ctx2.comm_finalize()
        assert ctx2.comm_info() == (0, 1)
    finally:
        ctx2.close()


def test_grouped_collectives_run_on_the_real_rccl_with_one_rank(ctx):
    """on a 1-GPU box the real librccl had only ever seen ncclGetUniqueId / CommInitRank / CommDestroy —
    the exchange and the degree all-reduce return early for one rank.  The test-only option `dist_force_self` keeps the calls:
    per level a grouped self ncclSend / ncclRecv (dist_collective 0) or an ncclBroadcast from root 0 (1) into a scratch buffer
    that REPLACES the rank's own frontier words, plus the ncclAllReduce of the degree vector at set-up — so symbol binding,
    group nesting and stream ordering execute on real RCCL, and the levels only match the oracle if the bytes arrived."""
    a = oracle.rmat_csr(13)
    for coll in (0, 1):
        ctx2 = engine.Context(0)
        try:
            ctx2.comm_init_rank(1, 0, ctx2.comm_unique_id())
            ctx2.set_option("dist_force_self", 1)
            ctx2.set_option("dist_collective", coll)
            assert ctx2.get_option("dist_force_self") == 1 and ctx2.get_option("dist_self_calls") == 0
            A = ctx2.mat_rmat(13)
            plan = engine.BfsPlan(ctx2, A, A.transpose(), 0, 1, splits=A.balanced_splits(1))
            levels = 0
            for src in (3, int(np.argmax(np.diff(a.rowptr)))):
                for force in (0, 1, 2):
                    plan.tune(force_direction=force)
                    engine.bfs_dist_run([plan], src)
                    lv, _ = plan.fetch()
                    ref = oracle.bfs(a, src, -1)[0]
                    np.testing.assert_array_equal(lv[:a.nrows], ref)
                    levels += int(ref.max())
            calls = ctx2.get_option("dist_self_calls")
            assert calls >= levels + 1, (calls, levels)     # one exchange per level that ran + the degree all-reduce
            plan.free()
            ctx2.comm_finalize()
        finally:
            ctx2.close()


def test_bench_gang_leg_on_one_device():
    """bench.py's second BFS-26 leg for N > 1 (one process driving every GPU, frontier words stored into the peers' bitmaps):
    the leg's own code — contexts, slabs, plans, fgpu_bfs_dist_run in peer mode, the time split — on a small graph with both
    "ranks" on the one GPU of a test box.  The BFS results of that mode are held to the oracle by
    test_in_library_dist_loop_matches_the_oracle; here the leg must run and report consistent numbers."""
    import types
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    args = types.SimpleNamespace(edge_factor=16, alpha=0.0, force_dir=0, opt=[])
    g = bench.bfs_gang_leg(engine, args, 16, 2, None, 6, 2, [], devices=[0, 0])
    assert g["ranks"] == 2 and g["TEPS"] > 0 and g["levels_per_search"] >= 2
    assert sum(g["slab_nnz"]) == g["edges"] and len(g["per_rank_ms_per_search"]) == 2
    a = oracle.rmat_csr(16)
    assert g["edges"] == a.nnz and g["vertices"] == a.nrows


@pytest.mark.parametrize("late", [0, 2, 3])
def test_gang_survives_a_peer_that_finishes_its_levels_late(late):
    """The single-process gang (fgpu_bfs_dist_run in peer mode: one context and one stream per rank, every rank's frontier
    words stored into every peer's bitmap, the next level ordered behind every source's `copied` event) with ONE rank
    finishing each of its levels 400 microseconds late (option dist_test_delay_us: a kernel spinning behind its level
    kernel): the peers' next levels must wait for the late rank's words, whichever rank it is — levels against the oracle for
    several roots, and the blind level budget must still end every search.  Four "ranks" share the one GPU of a test box."""
    scale, nranks = 15, 4
    a = oracle.rmat_csr(scale)
    n = a.nrows
    ctxs = [engine.Context(0) for _ in range(nranks)]
    plans, keep = [], []
    try:
        full0 = ctxs[0].mat_rmat(scale)
        splits = full0.balanced_splits(nranks)
        for r, c in enumerate(ctxs):
            full = full0 if r == 0 else c.mat_rmat(scale)
            A = full.col_slab(int(splits[r]), int(min(splits[r + 1], n)))
            full.free()
            At = A.transpose()
            keep += [A, At]
            plans.append(engine.BfsPlan(c, A, At, r, nranks, splits=splits))
        ctxs[late].set_option("dist_test_delay_us", 400)
        for c in ctxs:
            c.sync()
        deg = np.diff(a.rowptr)
        for src in [int(np.argmax(deg)), 7, int(np.nonzero(deg > 0)[0][-1]), 1234]:
            engine.bfs_dist_run(plans, src, -1, False)
            ref, _, ref_edges = oracle.bfs(a, src, -1)
            level = np.full(n, -1, dtype=np.int32)
            for r, p in enumerate(plans):
                lv, _ = p.fetch(want_parent=False)
                lo, hi = int(splits[r]), int(min(splits[r + 1], n))
                level[lo:hi] = lv[lo:hi]
            np.testing.assert_array_equal(level, ref)
            assert sum(p.stats()["edges_traversed"] for p in plans) == ref_edges
    finally:
        for p in plans:
            p.free()
        for m_ in keep:
            m_.free()
        for c in ctxs:
            c.close()


def test_bench_spawns_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (it used to exit with an error):
    bench.py re-executes itself under torch.distributed.run, one rank per GPU, and still prints ONE line.  Both ranks share
    the one GPU of a test box (FGPU_BENCH_ONE_DEVICE, gloo for the launcher's collectives)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FGPU_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scale", "18",
           "--no-cpu-baseline", "--no-pmc", "--no-varlen"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["parity"]["ok"] is True and d["value"] > 0


# ---- the multi-rank RCCL branch of the exchange, executed on one GPU through the loop-back library -------------------

_STUB_SCRIPT = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["FGPU_TEST_ROOT"])
import oracle
from falkordb_amd import engine

scale = 14
a = oracle.rmat_csr(scale)
n = a.nrows
deg = np.diff(a.rowptr)
report = []
for nranks, splits_kind, coll in [(2, "balanced", 0), (3, "balanced", 0), (3, "balanced", 1), (4, "empty", 0), (4, "empty", 1),
                                  (4, "balanced", 0)]:
    ctxs = [engine.Context(0) for _ in range(nranks)]          # one context per "rank", all on device 0
    engine.comm_init_all(ctxs)                                 # ncclCommInitAll through libfgpu.so
    assert [c.comm_info() for c in ctxs] == [(r, nranks) for r in range(nranks)]
    mats = [c.mat_rmat(scale) for c in ctxs]
    if splits_kind == "balanced":
        splits = mats[0].balanced_splits(nranks)
    else:                                                      # rank 1 owns nothing: counts[1] == 0 on every side
        splits = np.array([0, 8192, 8192, 12288, 16384], dtype=np.uint64)
    plans, keep = [], []
    for r, (c, A) in enumerate(zip(ctxs, mats)):
        c.set_option("dist_collective", coll)
        lo, hi = int(splits[r]), int(min(splits[r + 1], n))
        a_slab = A.col_slab(lo, max(hi, lo))
        at_slab = a_slab.transpose()
        keep += [a_slab, at_slab]
        plans.append(engine.BfsPlan(c, a_slab, at_slab, r, nranks, splits=splits))
    for force in (0, 1, 2):
        for p in plans:
            p.tune(force_direction=force)
        for src in [int(np.argmax(deg)), 5, int(np.nonzero(deg > 0)[0][-1])]:
            for max_level in (-1, 2):
                engine.bfs_dist_run(plans, src, max_level, want_parent=True)
                ref, _, ref_edges = oracle.bfs(a, src, max_level)
                level = np.full(n, -1, dtype=np.int32)
                parent = np.full(n, -1, dtype=np.int64)
                for r, p in enumerate(plans):
                    lv, par = p.fetch(want_parent=True)
                    lo, hi = int(splits[r]), int(min(splits[r + 1], n))
                    level[lo:hi] = lv[lo:hi]
                    parent[lo:hi] = par[lo:hi]
                assert np.array_equal(level, ref), (nranks, splits_kind, coll, force, src, max_level)
                reached = np.nonzero(ref > 0)[0]
                assert (ref[parent[reached]] + 1 == ref[reached]).all() and parent[src] == src
                if max_level < 0:
                    st = [p.stats() for p in plans]
                    assert sum(x["reached"] for x in st) == int((ref >= 0).sum())
                    assert sum(x["edges_traversed"] for x in st) == ref_edges      # needs the all-reduced global degrees
    stub = ctypes.CDLL(os.environ["FGPU_RCCL_LIB"])
    cnt = (ctypes.c_long * 4)()
    stub.fgpu_stub_rccl_counters(cnt)
    report.append({"nranks": nranks, "splits": splits_kind, "collective": coll, "counters": list(cnt)})
    for p in plans:
        p.free()
    for c in ctxs:
        c.comm_finalize()
        c.close()
print("STUB_REPORT " + json.dumps(report))
"""


def test_in_library_dist_loop_over_the_multi_rank_communicator_branch(tmp_path):
    """the `nranks > 1` branch of the exchange had never executed anywhere (1-GPU boxes).  libfgpu.so binds
    its eleven RCCL entry points with dlopen; FGPU_RCCL_LIB points it at tests/stub_rccl (an in-process loop-back:
    ranks = communicators of one process on one device), and fgpu_comm_init_all + fgpu_bfs_dist_run then run the real
    multi-rank code — group nesting for the gang, grouped ncclSend / ncclRecv per peer pair (dist_collective 0) and
    ncclBroadcast per rank (1), the ncclAllReduce of the degree vectors, uneven and EMPTY slabs — against the oracle,
    for 2 / 3 / 4 ranks and every direction mode.  Runs in a child process: the binding is once per process, and this
    one already holds the real RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = os.path.join(root, "tests", "stub_rccl", "libstub_rccl.so")
    if not os.path.exists(stub):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                        os.path.join(root, "tests", "stub_rccl", "stub_rccl.hip"), "-o", stub], check=True)
    env = dict(os.environ, FGPU_RCCL_LIB=stub, FGPU_TEST_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _STUB_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("STUB_REPORT ")][-1]
    rep = json.loads(line[len("STUB_REPORT "):])
    assert len(rep) == 6
    sends = [x["counters"][0] for x in rep]
    bcasts = [x["counters"][1] for x in rep]
    allred = [x["counters"][2] for x in rep]
    assert sends[0] > 0 and sends[1] > sends[0]                 # send/recv pairs ran (dist_collective 0)
    assert bcasts[2] > bcasts[1] and bcasts[4] > bcasts[3]      # broadcasts ran (dist_collective 1)
    assert allred == sorted(allred) and allred[0] >= 1 and allred[-1] >= 6   # one degree all-reduce per partition


def test_bench_line_with_two_ranks_on_one_device(tmp_path):
    """The N > 1 control flow of bench.py — source batches sharded round-robin over the ranks, adjacency replicated, per-rank
    timed loops between fences, flops summed and time max-reduced over the ranks, rank 0 printing the line — launched the way
    the driver launches it (torch.distributed.run, one process per rank) with both ranks on the one GPU of a test box and the
    launcher's collectives over gloo (FGPU_BENCH_ONE_DEVICE: RCCL refuses two ranks on one device, so the RCCL BFS leg is
    skipped).  Not a measurement: it pins that the line of a multi-rank run parses, names 2 GPUs, and carries twice the
    traversed edges of the same steps on one rank's share."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FGPU_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--scale", "18",
           "--no-cpu-baseline", "--no-pmc", "--no-varlen"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["unit"] == "TEPS"
    assert d["metric"].startswith("traversed edges/sec (TEPS) on k-hop MATCH")
    assert d["parity"]["ok"] is True and d["value"] > 0
    assert "2 GPUs" in d["config"]["parallelism"]
