"""Tier 2 of the boundary: falkordb_amd/lib/libgraphblas.so exports the GrB_* / GxB_* symbols the
reference's matrix.rs binds.  tests/shim/replay_matrix_rs.c — a C program written ONLY against declarations transcribed
from the reference's bindgen output (tests/shim/graphblas_subset.h cites mod.rs line by line) — replays Matrix::new /
build / delta_lmxm / Iter call for call; its output must be the oracle's delta_lmxm chain (oracle.expand_omp, the
restatement of matrix.rs:1317-1402 pinned by tests/test_oracle_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "falkordb_amd", "lib")


@pytest.fixture(scope="module")
def replay_exe(tmp_path_factory):
    from falkordb_amd import build as fb
    fb.build_shim()
    exe = str(tmp_path_factory.mktemp("shim") / "replay_matrix_rs")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "tests", "shim"),
                    os.path.join(ROOT, "tests", "shim", "replay_matrix_rs.c"), "-o", exe, "-L" + LIBDIR, "-lgraphblas",
                    "-Wl,-rpath," + LIBDIR], check=True)
    return exe


def _write(path, n, nsrc, nhops, valued, m, dp, dm, src, vals_of):
    with open(path, "w") as f:
        f.write(f"{n} {nsrc} {nhops} {1 if valued else 0}\n")
        for layer, with_vals in ((m, valued), (dp, valued), (dm, False)):
            r, c = layer.pairs()
            if layer is m and len(r) > 50:      # duplicates in the COO stream: build collapses them (matrix.rs:1686-1695)
                r, c = np.concatenate([r, r[:50]]), np.concatenate([c, c[:50]])
            f.write(f"{len(r)}\n")
            for i, j in zip(r.tolist(), c.tolist()):
                f.write(f"{i} {j} {vals_of(i, j)}\n" if with_vals else f"{i} {j}\n")
        f.write(" ".join(str(int(s)) for s in src) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("valued", [False, True])
@pytest.mark.parametrize("dirty", [False, True])
@pytest.mark.parametrize("nhops", [1, 3])
def test_matrix_rs_call_sequences_through_the_graphblas_abi_match_the_oracle(replay_exe, tmp_path, valued, dirty, nhops):
    scale = 11
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rng = np.random.default_rng(17 + nhops)
    rows, cols = a.pairs()
    if dirty:
        pick = rng.choice(len(rows), 300, replace=False)
        dm = oracle.build_csr(n, n, rows[pick], cols[pick])
        raw = oracle.build_csr(n, n, rng.integers(0, n, 300).astype(np.uint64), rng.integers(0, n, 300).astype(np.uint64))
        dp = oracle.merge(raw, None, a)                    # dp ∩ m = ∅ (versioned_matrix.rs:214-235)
    else:
        dm = dp = oracle.empty(n, n)
    src = rng.integers(0, n, 40).astype(np.uint64)
    vals_of = lambda i, j: (i * 1315423911 + j * 2654435761) & 0x7FFFFFFFFFFFFFFF
    inp = tmp_path / "case.txt"
    _write(inp, n, len(src), nhops, valued, a, dp, dm, src, vals_of)
    env = dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([replay_exe, str(inp)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert lines[0].startswith("nvals ")
    nvals = int(lines[0].split()[1])
    got = [tuple(int(x) for x in l.split()) for l in lines[1:1 + nvals]]
    c, _, _ = oracle.expand_omp(src, [(a, dp if dirty else None, dm if dirty else None)] * nhops)
    cr, cc = c.pairs()
    assert nvals == c.nnz
    assert got == list(zip(cr.tolist(), cc.tolist()))       # ascending (row, col): the iterator's order (matrix.rs:1572-1605)
    probes = [l.split() for l in lines[1 + nvals:] if l.startswith("probe")]
    # Encode<19> / Decode<19> through GxB_Container + GxB_Vector_load / _unload (matrix.rs:428-546, vector.rs:241-420)
    cont = {l.split()[1]: l.split()[2:] for l in lines if l.startswith("container ")}
    assert int(cont["m"][0]) == a.nnz and cont["m"][3] == "1", cont
    assert cont["m"][2] == ("0" if valued else "1")                        # iso: BOOL matrices are patterns, UINT64 ones are not
    assert int(cont["dm"][0]) == dm.nnz and cont["dm"][3] == "1", cont
    # the multi-edge id list (tensor.rs:1111-1120) and a Vector<u64> through serialize / deserialize / the vector iterator
    assert [l for l in lines if l.startswith("ids ")] == [f"ids 1 5 500 501 {1 << 40}"]
    assert [l for l in lines if l.startswith("u64vec")] == [f"u64vec 3:{1 << 63} 9:77"]
    mem = [l.split() for l in lines if l.startswith("mem ")][0]
    assert int(mem[1]) >= 4 * (n + 1 + a.nnz) and mem[3] == ("0" if valued else "1")
    first = (0, 0) in a.to_set()
    assert int(probes[0][1]) == (0 if first else 1)         # GrB_SUCCESS / GrB_NO_VALUE
    if first and valued:
        assert int(probes[0][2]) == vals_of(0, 0)
    assert int(probes[1][1]) == (0 if (n - 1, n - 1) in a.to_set() else 1)
    assert int(probes[2][1]) == -4                          # GrB_INVALID_INDEX for a row past the end


@pytest.fixture(scope="module")
def replay_algo_exe(tmp_path_factory):
    from falkordb_amd import build as fb
    fb.build_shim()
    exe = str(tmp_path_factory.mktemp("shim") / "replay_algo_rs")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "tests", "shim"),
                    os.path.join(ROOT, "tests", "shim", "replay_algo_rs.c"), "-o", exe, "-L" + LIBDIR, "-llagraphx", "-llagraph",
                    "-lgraphblas", "-Wl,-rpath," + LIBDIR], check=True)                 # the three names build.rs:50-52 links
    return exe


def _blocks(lines):
    """replay_algo_rs output -> list of (header words, rows of ints / floats)."""
    out = []
    for l in lines:
        w = l.split()
        if not w:
            continue
        if w[0][0].isalpha():
            out.append((w, []))
        else:
            out[-1][1].append(w)
    return out


@pytest.mark.gpu
def test_algo_rs_call_sequences_through_the_lagraph_abi_match_the_oracle(replay_algo_exe, tmp_path):
    """algo.BFS and algo.pageRank exactly as algo_procedures.rs:1060-1165 / :718-760 issue them — LAGraph_New over a
    borrowed adjacency, LAGr_BreadthFirstSearch_Extended, GrB_Vector_extractTuples_INT64, LAGraph_Delete; LAGraph_Cached_AT
    + _OutDegree, LAGr_PageRank, extractTuples_FP64 — against oracle.bfs (the LAGr_BreadthFirstSearch_Extended contract,
    pinned by tests/test_oracle_golden.py) and oracle/pagerank.py."""
    from oracle import pagerank as opr
    scale = 11
    a = oracle.rmat_csr(scale)
    n = a.nrows
    rows, cols = a.pairs()
    deg = np.diff(a.rowptr.astype(np.int64))
    src = int(np.argmax(deg))
    lone = int(np.nonzero(deg == 0)[0][0])                    # a vertex without out-edges: reaches only itself
    cases = [(src, -1, 1), (src, 2, 0), (src, 1, 1), (lone, -1, 1), (5, 0, 0)]
    grown = n + 7                                             # deleted ids stay as isolated vertices (:722-723)
    inp = tmp_path / "algo.txt"
    with open(inp, "w") as f:
        f.write(f"{n} {len(rows)}\n")
        for i, j in zip(rows.tolist(), cols.tolist()):
            f.write(f"{i} {j}\n")
        for c in cases:
            f.write("bfs %d %d %d\n" % c)
        f.write(f"pagerank {grown}\nerrors\n")
        f.write("bfs %d %d %d\n" % cases[0])                  # again after the PageRank run: the borrowed matrix is intact
    env = dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([replay_algo_exe, str(inp)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = _blocks(r.stdout.split("\n"))
    edge_set = a.to_set()
    bi = 0
    for (s0, max_level, want) in cases + [cases[0]]:
        head, body = blocks[bi]
        bi += 1
        assert head[:4] == ["bfs", str(s0), str(max_level), str(want)]
        level_ref, _, _ = oracle.bfs(a, s0, max_level, True)
        reached = np.nonzero(level_ref >= 0)[0]
        assert int(head[5]) == len(reached)
        got_idx = np.array([int(w[0]) for w in body], dtype=np.int64)
        got_lvl = np.array([int(w[1]) for w in body], dtype=np.int64)
        assert (got_idx == reached).all()                      # ascending indices, exactly the reached set
        assert (got_lvl == level_ref[reached]).all()
        if want:
            head, body = blocks[bi]
            bi += 1
            assert head == ["parent", str(len(reached))]
            pidx = np.array([int(w[0]) for w in body], dtype=np.int64)
            par = np.array([int(w[1]) for w in body], dtype=np.int64)
            assert (pidx == reached).all()
            for v, p in zip(pidx.tolist(), par.tolist()):      # any valid BFS tree (LAGraph's parent is one of them)
                if v == s0:
                    assert p == s0
                else:
                    assert level_ref[p] == level_ref[v] - 1 and (p, v) in edge_set
        if (s0, max_level, want) == cases[-1] and bi < len(blocks) and blocks[bi][0][0] == "pagerank":
            head, body = blocks[bi]
            bi += 1
            big = oracle.build_csr(grown, grown, rows, cols)
            ref, it_ref = opr.pagerank(big)
            assert int(head[1]) == grown and int(head[5]) == grown         # a full vector: one score per vertex
            assert int(head[7]) == int((deg > 0).sum())                    # out_degree stores the non-zero degrees only
            it = int(head[3])
            got = np.array([float(w[1]) for w in body], dtype=np.float32)
            assert [int(w[0]) for w in body] == list(range(grown))
            assert abs(it - it_ref) <= 1
            if it == it_ref:
                np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)    # north_star: 1e-6 rel for PLUS_TIMES float semirings
            else:
                assert float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).sum()) <= 2e-4
            errs = {}
            while blocks[bi][0][0] == "errors":
                errs[blocks[bi][0][1]] = blocks[bi][0][2:]
                bi += 1
            assert errs["not_cached"] == ["-1003", "1"]                    # LAGRAPH_NOT_CACHED (lagraph_bindings.rs:26)
            assert errs["bad_source"] == ["-4", "1"]                       # GrB_INVALID_INDEX
            assert errs["no_convergence"] == ["-1005", "1", "2"]           # LAGRAPH_CONVERGENCE_FAILURE after itermax = 2, from ONE run
            assert errs["off_path"] == ["-8", "1", "message"]              # GrB_NOT_IMPLEMENTED, loudly
            assert errs["null_graph"] == ["-2"] and errs["no_matrix"] == ["-1000"]
    assert blocks[bi][0] == ["adjacency", str(a.nnz)]
    assert blocks[bi + 1][0] == ["allocator_blocks", "0"]      # nothing of the caller's allocator is left behind


def test_replay_programs_link_against_the_three_libraries(tmp_path):
    """Link completeness at LINK time, without a GPU: both replay programs — written against the transcribed bindgen
    declarations only — link with `-z defs` semantics of an executable (every symbol they reference must be defined by
    libgraphblas.so / liblagraph.so / liblagraphx.so), using the three library names build.rs:50-52 links."""
    from falkordb_amd import build as fb
    fb.build_shim()
    for src, libs in (("replay_matrix_rs.c", ["-lgraphblas"]), ("replay_algo_rs.c", ["-llagraphx", "-llagraph", "-lgraphblas"])):
        exe = str(tmp_path / src[:-2])
        r = subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror=implicit-function-declaration",
                            "-I" + os.path.join(ROOT, "tests", "shim"), os.path.join(ROOT, "tests", "shim", src), "-o", exe,
                            "-L" + LIBDIR] + libs + ["-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.exists(exe)


def test_shim_exports_every_symbol_the_wrapper_imports():
    """Link completeness, checked mechanically: tests/golden/shim_symbols.json is GENERATED (tests/golden/
    make_shim_symbols.py, run in the build container) from the `use super::{...}` import lists of the reference's
    matrix.rs:79-102, vector.rs:43-60, tensor.rs and versioned_matrix.rs, keeping the names bindgen declares as extern
    functions / statics — every one of them must be a defined dynamic symbol of libgraphblas.so."""
    import json
    from falkordb_amd import build as fb
    so = fb.build_shim()
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    have = {l.split()[-1] for l in out.splitlines() if l.strip()}
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "shim_symbols.json")))
    assert len(want["functions"]) >= 60 and len(want["globals"]) >= 35
    for must in ("GrB_Matrix_build_BOOL", "GxB_Container_new", "GxB_load_Matrix_from_Container", "GxB_unload_Matrix_into_Container",
                 "GxB_Matrix_memoryUsage", "GxB_Matrix_fprint", "GrB_Vector_new", "GxB_Vector_serialize", "GxB_Vector_load"):
        assert must in want["functions"], must               # (the generator still sees the lists it was written for)
    missing = [s for s in want["functions"] + want["globals"] if s not in have]
    assert not missing, missing
    # the LAGraph half: every LAGraph_* / LAGr_* entry point matrix.rs and algo_procedures.rs call is defined by
    # liblagraph.so / liblagraphx.so (split as the reference's two binding files split them), and the GraphBLAS calls
    # algo.BFS / algo.pageRank make on top of the wrapper's own list are defined by libgraphblas.so
    for lib, key in (("liblagraph.so", "lagraph"), ("liblagraphx.so", "lagraphx")):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, lib)], capture_output=True, text=True, check=True).stdout
        got = {l.split()[-1] for l in out.splitlines() if l.strip()}
        assert want[key], key
        assert not [s for s in want[key] if s not in got], (lib, [s for s in want[key] if s not in got])
    assert "LAGr_BreadthFirstSearch_Extended" in want["lagraphx"] and "LAGr_PageRank" in want["lagraph"]
    assert not [s for s in want["algo_bfs_pagerank_graphblas"] if s not in have]
