"""CPU: the C++ host layer (libfalkor_host.so) builds, exports every symbol include/falkor_host.h declares,
its pure-integer pieces (fold policy, compound key) match the reference's pins, and — like the device
library under it — it refuses to start without a HIP device (no CPU fallback)."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "rust_unit_pins.json")))


def test_host_library_builds_and_exports_every_declared_symbol():
    from falkordb_amd import build, host
    lib_path = build.build_host()
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (fh_[a-z0-9_]+)", out))
    declared = host.declared_symbols()
    assert len(declared) > 50
    missing = [d for d in declared if d not in exported]
    assert not missing, f"declared in falkor_host.h but not exported: {missing}"
    host.load()


def test_host_layer_reaches_the_engine_only_through_the_c_abi():
    # the host layer is what the Rust side would be: it may include fgpu.h and nothing of csrc/
    hdir = os.path.join(ROOT, "falkordb_amd", "host")
    for f in os.listdir(hdir):
        text = open(os.path.join(hdir, f)).read()
        assert "csrc/" not in text and "common.hpp" not in text and "hip/hip_runtime" not in text, f
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "falkordb_amd", "lib",
                                                                        "libfalkor_host.so")],
                         capture_output=True, text=True, check=True).stdout
    used = set(re.findall(r" U (fgpu_[a-z0-9_]+)", out))
    # (expand_batch takes the chain's result as device-built columns since round 5 — fgpu_expand_pairs32 since round 6: the
    # destinations as the device's 32-bit node ids — or fgpu_expand_probe when every row has a bound destination; algo_bfs
    # fetches into pinned blocks)
    assert {"fgpu_init", "fgpu_expand_pairs32", "fgpu_expand_probe", "fgpu_host_alloc",
            "fgpu_bfs", "fgpu_mat_merge", "fgpu_mat_probe", "fgpu_delta_lmxm"} <= used
    assert not re.search(r" U hip[A-Z]", out), "host layer must not call HIP directly"


def _threshold(k, tx):
    import math
    t = math.isqrt(k * tx)
    return t if t * t >= k * tx else t + 1


def test_fold_thresholds_match_the_reference_pins():     # versioned_matrix.rs:1278-1330
    from falkordb_amd import host
    p = PINS["fold_thresholds"]
    HUGE = ((1 << 64) - 1) // 4
    assert p["threshold_read_tx1"] == 287 and p["threshold_write_tx1"] == 4528 and p["threshold_read_tx100"] == 2864
    for base in (1_000_000, 10_000_000, 100_000_000, HUGE):
        assert not host.should_fold_read(286, 1, base)
        assert host.should_fold_read(287, 1, base)
    assert not host.should_fold(4527, 1, HUGE) and host.should_fold(4528, 1, HUGE)
    for tx in (1, 10, 100, 1000):
        d = _threshold(p["READ_FOLD_K"], tx)
        assert not host.should_fold_read(d - 1, tx, HUGE) and host.should_fold_read(d, tx, HUGE)
    assert not host.should_fold_read(2863, 100, HUGE) and host.should_fold_read(2864, 100, HUGE)
    U = (1 << 64) - 1
    assert host.should_fold(512, U, 1024) and host.should_fold_read(512, U, 1024)          # escape hatch
    assert not host.should_fold(p["MIN_FOLD_DELTA"] - 1, 1, 0)                              # floors
    assert not host.should_fold_read(p["MIN_FOLD_DELTA"] - 1, 1, 0)
    assert not host.should_fold(U, 0, 1024) and not host.should_fold_read(U, 0, 1024)
    assert host.delta_dominates_base(256, 512) and not host.delta_dominates_base(255, 0)
    assert not host.delta_dominates_base(256, 513)


def test_fold_policy_agrees_with_the_oracle_on_a_grid():
    from falkordb_amd import host
    from oracle import model
    vals = [0, 1, 255, 256, 257, 286, 287, 1000, 4527, 4528, 10**6, 10**9, (1 << 63), (1 << 64) - 1]
    for d in vals:
        for tx in (0, 1, 7, 100, 10**6, (1 << 64) - 1):
            for base in (0, 1, 511, 512, 10**7, (1 << 64) - 1):
                assert host.should_fold(d, tx, base) == model.should_fold(d, tx, base), (d, tx, base)
                assert host.should_fold_read(d, tx, base) == model.should_fold_read(d, tx, base), (d, tx, base)
                assert host.delta_dominates_base(d, base) == model.delta_dominates_base(d, base)


def test_compound_key():                                 # tensor.rs:154-163
    from falkordb_amd import host
    assert host.compound_key(3, 5) == (3 << 32) | 5
    assert host.compound_key((1 << 32) - 1, (1 << 32) - 1) == (1 << 64) - 1
    with pytest.raises(host.HostError):
        host.compound_key(1 << 32, 0)
    with pytest.raises(host.HostError):
        host.compound_key(0, 1 << 32)


def test_cond_traverse_eligibility_rule():               # cond_traverse.rs:308-316
    from falkordb_amd import host
    L = host.load()
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])])) == 1
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])], emit=True)) == 0
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])], bidir=True)) == 0
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])], siblings=True)) == 0
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])], attrs=True)) == 0
    # a fused chain makes inline attrs irrelevant (they can only sit on the fused-away intermediates)
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], []), ([], [])], attrs=True)) == 1
    # `transposed` does not enter the rule (:308-316 test the pattern's shape): what makes the reference serve those per row is
    # the unbound matrix source at run time (:556-568); this engine's expand_batch takes them over the transposed layers
    assert L.fh_cond_traverse_eligible(host.cond_spec(hops=[(["R"], [])], transposed=True)) == 1
    assert b"transposed=1" in host.cond_spec(hops=[(["R"], [])], transposed=True)
    assert b"transposed=0" in host.cond_spec(hops=[(["R"], [])])


def test_no_cpu_fallback_host_init_fails_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from falkordb_amd import host
    with pytest.raises(host.HostError) as e:
        host.Context(0)
    assert e.value.code == -7002    # FGPU_DEVICE


# ---- planner slice: fuse_anonymous_traverse (planner/optimizer/fuse_anonymous_traverse.rs:83-284) ----------------
def _node(alias, labels=(), attrs=False):
    return {"alias": alias, "labels": list(labels), "attrs": attrs}


def _rel(alias, frm, to, types=("KNOWS",), **kw):
    return {"alias": alias, "from": frm, "to": to, "types": list(types), "bidirectional": kw.get("bidirectional", False),
            "var_len": kw.get("var_len", False), "attrs": kw.get("attrs", False)}


def _ct(i, parent, rel, **kw):
    return {"id": i, "parent": parent, "kind": "CT", "rel": rel, "emit": kw.get("emit", False),
            "transposed": kw.get("transposed", False), "optional": kw.get("optional", False), "bind": kw.get("bind", True),
            "siblings": list(kw.get("siblings", [])), "chain": list(kw.get("chain", []))}


def _x(i, parent, name, refs=()):
    return {"id": i, "parent": parent, "kind": "X", "name": name, "refs": list(refs)}


def _three_hop(override=None):
    """MATCH (a:P)-->()-->()-->(c:Q) RETURN count(c): Project / CT(_b2->c) / CT(_b1->_b2) / CT(a->_b1) / scan"""
    a, b1, b2, c = _node("a", ["P"]), _node("_anon_1"), _node("_anon_2"), _node("c", ["Q"])
    ops = [_x(0, -1, "Aggregate", ["c"]),
           _ct(1, 0, _rel("_anon_e3", b2, c)),
           _ct(2, 1, _rel("_anon_e2", b1, b2)),
           _ct(3, 2, _rel("_anon_e1", a, b1)),
           _x(4, 3, "NodeByLabelScan", [])]
    for (op_id, path), value in (override or {}).items():
        tgt = ops[op_id]
        for k in path[:-1]:
            tgt = tgt[k]
        tgt[path[-1]] = value
    return ops


def _canon(ops):
    return sorted((json.dumps(o, sort_keys=True) for o in ops))


def test_fuse_anonymous_traverse_collapses_a_three_hop_chain():
    from falkordb_amd import host
    from oracle import model
    ops = _three_hop()
    got, spec = host.plan_fuse(ops, lower_id=1)
    want = model.fuse_anonymous_traverse(ops)
    assert _canon(got) == _canon(want)
    cts = [o for o in got if o["kind"] == "CT"]
    assert len(cts) == 1 and cts[0]["id"] == 1 and cts[0]["parent"] == 0
    ct = cts[0]
    # entry hop first, then the chain in traversal order (:236-241); the scan hangs under the merged op (:266-271)
    assert ct["rel"]["alias"] == "_anon_e1" and [r["alias"] for r in ct["chain"]] == ["_anon_e2", "_anon_e3"]
    assert [o for o in got if o["id"] == 4][0]["parent"] == 1
    assert not ct["transposed"] and not ct["optional"] and ct["bind"]
    # the runtime operator: source label P, three hops, destination label Q on the last hop only
    assert spec == b"src=P;hop=KNOWS|;hop=KNOWS|;hop=KNOWS|Q;optional=0;bind=0;emit=0;bidir=0;siblings=0;attrs=0"
    L = host.load()
    assert L.fh_cond_traverse_eligible(spec) == 1


@pytest.mark.parametrize("name,override,fused_pairs", [
    ("optional hop", {(2, ("optional",)): True}, 0),
    ("transposed hop", {(2, ("transposed",)): True}, 0),
    ("named edge", {(2, ("rel", "alias")): "e"}, 0),
    ("emitted edge", {(2, ("emit",)): True}, 0),
    ("sibling edges", {(2, ("siblings",)): ["_anon_e9"]}, 0),
    ("bidirectional hop", {(2, ("rel", "bidirectional")): True}, 0),
    ("variable length hop", {(2, ("rel", "var_len")): True}, 0),
    ("edge attributes", {(2, ("rel", "attrs")): True}, 0),
    ("bind_relationship off", {(2, ("bind",)): False}, 0),
    # the middle hop spoils both pairs; spoiling only an END hop leaves the other pair fusable:
    ("named last edge", {(1, ("rel", "alias")): "e"}, 1),
    ("optional first hop", {(3, ("optional",)): True}, 1),
])
def test_fuse_anonymous_traverse_conditions(name, override, fused_pairs):
    from falkordb_amd import host
    from oracle import model
    ops = _three_hop(override)
    got, _ = host.plan_fuse(ops)
    want = model.fuse_anonymous_traverse(ops)
    assert _canon(got) == _canon(want), name
    assert sum(o["kind"] == "CT" for o in got) == 3 - fused_pairs, name


@pytest.mark.parametrize("name,mutate,n_ct", [
    ("labelled intermediate", lambda ops: ops[1]["rel"]["from"]["labels"].append("P") or ops[2]["rel"]["to"]["labels"].append("P"), 2),
    ("named intermediate", lambda ops: [ops[1]["rel"]["from"].update(alias="b"), ops[2]["rel"]["to"].update(alias="b")], 2),
    ("intermediate with attributes", lambda ops: [ops[1]["rel"]["from"].update(attrs=True), ops[2]["rel"]["to"].update(attrs=True)], 2),
    ("intermediate referenced above", lambda ops: ops[0]["refs"].append("_anon_2"), 2),
    ("hops do not share the intermediate",
     lambda ops: ops[1]["rel"].update({"from": dict(ops[1]["rel"]["from"], alias="_anon_9")}), 2),
])
def test_fuse_anonymous_traverse_intermediate_rules(name, mutate, n_ct):
    from falkordb_amd import host
    from oracle import model
    ops = _three_hop()
    mutate(ops)
    got, _ = host.plan_fuse(ops)
    want = model.fuse_anonymous_traverse(ops)
    assert _canon(got) == _canon(want), name
    assert sum(o["kind"] == "CT" for o in got) == n_ct, name


def test_fuse_anonymous_traverse_needs_an_only_child_and_keeps_branches():
    from falkordb_amd import host
    from oracle import model
    # a CondTraverse with two children (e.g. under an Apply) is left alone (:198-200); a fusable pair elsewhere in the
    # tree is still merged, and a filter between the hops blocks the pair it separates (the child is not a CT)
    a, b1, b2, c = _node("a"), _node("_anon_1"), _node("_anon_2"), _node("c")
    ops = [_x(0, -1, "Project", ["a", "c"]),
           _ct(1, 0, _rel("_anon_e2", b1, c)),
           _ct(2, 1, _rel("_anon_e1", a, b1)),
           _x(3, 2, "AllNodeScan"),
           _x(4, 1, "Argument")]                       # second child of op 1
    got, _ = host.plan_fuse(ops)
    assert _canon(got) == _canon(model.fuse_anonymous_traverse(ops)) == _canon(ops)
    ops2 = [_x(0, -1, "Project", ["c"]),
            _ct(1, 0, _rel("_anon_e3", b2, c)),
            _x(2, 1, "Filter", ["a"]),
            _ct(3, 2, _rel("_anon_e2", b1, b2)),
            _ct(4, 3, _rel("_anon_e1", a, b1)),
            _x(5, 4, "AllNodeScan")]
    got2, spec = host.plan_fuse(ops2, lower_id=3)
    want2 = model.fuse_anonymous_traverse(ops2)
    assert _canon(got2) == _canon(want2)
    assert sorted(o["id"] for o in got2 if o["kind"] == "CT") == [1, 3]
    assert spec.startswith(b"src=;hop=KNOWS|;hop=KNOWS|;")


# ---- v19 matrix payload: GxB_Container_struct bytes + five unload-form vectors (matrix.rs:428-546, vector.rs:241-309)
def _frame_u(v):
    import struct
    return struct.pack("<Q", v)


def _frame_buf(b):
    return _frame_u(len(b)) + bytes(b)


def _vec(arr, type_name):
    """Vector<bool>::encode: buffer(array), buffer(type name + NUL), unsigned(n_entries), unsigned(n_bytes), signed(handling)"""
    import numpy as np
    raw = np.ascontiguousarray(arr).tobytes()
    return _frame_buf(raw) + _frame_buf(type_name.encode() + b"\0") + _frame_u(len(arr)) + _frame_u(len(raw)) + _frame_u(0)


def _container(nrows, ncols, p, i, x=None, h=None, idx_bits=64, fmt=None, orientation=0, iso=None, jumbled=False):
    """The payload an encoder following graphblas/mod.rs:14165-14188 + matrix.rs:506-546 writes, built independently
    of the library: struct offsets nrows 0, ncols 8, nrows_nonempty 16, ncols_nonempty 24, nvals 32, format 128,
    orientation 132, iso 448, jumbled 449, size 608."""
    import struct
    import numpy as np
    st = bytearray(608)
    nvals = int(p[-1]) if len(p) else 0
    struct.pack_into("<QQqqQ", st, 0, nrows, ncols, -1, -1, nvals)
    struct.pack_into("<ii", st, 128, fmt if fmt is not None else (1 if h is not None else 2), orientation)
    st[448] = 1 if (iso if iso is not None else x is None) else 0
    st[449] = 1 if jumbled else 0
    it = np.uint32 if idx_bits == 32 else np.uint64
    tn = "GrB_UINT32" if idx_bits == 32 else "GrB_UINT64"
    xs = _vec(np.array([1] if nvals else [], dtype=np.uint8), "GrB_BOOL") if x is None else _vec(np.asarray(x, dtype=np.uint64), "GrB_UINT64")
    hs = _vec(np.asarray(h if h is not None else [], dtype=it), tn)
    return _frame_buf(st) + xs + hs + _vec(np.asarray(p, dtype=it), tn) + _vec(np.asarray(i, dtype=it), tn) + \
        _vec(np.array([], dtype=np.int8), "GrB_INT8")


@pytest.mark.parametrize("idx_bits", [32, 64])
def test_container_payload_parse(idx_bits):
    import numpy as np
    from falkordb_amd import host
    # sparse 5 x 7 iso-bool: rows 0: {1,4}, 2: {0}, 4: {2,3,6}
    d = host.container_parse(_container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6], idx_bits=idx_bits) + b"trailing")
    assert (d["nrows"], d["ncols"], d["nvals"], d["hyper"], d["valued"]) == (5, 7, 6, False, False)
    assert d["p"].tolist() == [0, 2, 2, 3, 3, 6] and d["i"].tolist() == [1, 4, 0, 2, 3, 6]
    assert d["consumed"] == len(_container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6], idx_bits=idx_bits))
    # hypersparse 10^9 x 10^9 UINT64: stored rows 3 and 999999999
    big = 10 ** 9
    d = host.container_parse(_container(big, big, [0, 1, 3], [5, 0, big - 1], x=[11, 22, 2 ** 63 + 5], h=[3, big - 1],
                                        idx_bits=idx_bits))
    assert d["hyper"] and d["valued"] and d["h"].tolist() == [3, big - 1]
    assert d["p"].tolist() == [0, 1, 3] and d["i"].tolist() == [5, 0, big - 1] and d["x"].tolist() == [11, 22, 2 ** 63 + 5]
    # iso UINT64 (one stored value for every entry) and the empty matrix
    d = host.container_parse(_container(3, 3, [0, 1, 2, 2], [2, 0], x=[9], iso=True, idx_bits=idx_bits))
    assert d["x"].tolist() == [9, 9]
    d = host.container_parse(_container(4, 4, [0, 0, 0, 0, 0], [], idx_bits=idx_bits))
    assert d["nvals"] == 0 and d["p"].tolist() == [0] * 5


@pytest.mark.parametrize("bad,why", [
    (lambda: _container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6])[:300], "truncated"),
    (lambda: _container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 9]), "column out of range"),
    (lambda: _container(5, 7, [0, 2, 1, 3, 3, 6], [1, 4, 0, 2, 3, 6]), "pointers decrease"),
    (lambda: _container(5, 7, [0, 2, 2, 3, 3, 5], [1, 4, 0, 2, 3, 6], fmt=4), "bitmap format"),
    (lambda: _container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6], orientation=1), "column major"),
    (lambda: _container(5, 7, [0, 2, 2, 3, 3, 6], [1, 4, 0, 2, 3, 6], jumbled=True), "jumbled"),
    (lambda: _container(5, 7, [0, 2, 3], [1, 4, 0], h=[4, 2]), "hyper list not ascending"),
    (lambda: _container(5, 7, [0, 2, 2, 3], [1, 4, 0]), "sparse with nvec != nrows"),
    (lambda: _frame_buf(b"x" * 100), "struct too small"),
])
def test_container_payload_rejects_malformed_input(bad, why):
    from falkordb_amd import host
    with pytest.raises(host.HostError):
        host.container_parse(bad())


def test_container_payload_parse_random_matrices():
    """Property test (hypothesis): any sorted-unique CSR, sparse or hypersparse, 32- or 64-bit indices, iso BOOL or
    UINT64 values, survives build-by-hand -> parse unchanged, and the parser consumes exactly the payload."""
    import numpy as np
    from hypothesis import given, settings, strategies as st
    from falkordb_amd import host

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 40), st.integers(1, 40), st.integers(0, 2 ** 32 - 1), st.booleans(), st.booleans(),
           st.sampled_from([32, 64]))
    def check(nrows, ncols, seed, hyper, valued, bits):
        rng = np.random.default_rng(seed)
        dense = rng.random((nrows, ncols)) < 0.15
        if hyper:
            dense[rng.random(nrows) < 0.6] = False              # most rows empty
        rows, cols = np.nonzero(dense)
        counts = np.bincount(rows, minlength=nrows)
        if hyper:
            stored = np.nonzero(counts)[0]
            p = np.concatenate([[0], np.cumsum(counts[stored])])
            h = stored
        else:
            p = np.concatenate([[0], np.cumsum(counts)])
            h = None
        x = rng.integers(0, 2 ** 63, len(cols)) if valued else None
        payload = _container(nrows, ncols, p, cols, x=x, h=h, idx_bits=bits)
        d = host.container_parse(payload + b"\\x00\\x01")
        assert d["consumed"] == len(payload)
        assert (d["nrows"], d["ncols"], d["nvals"], d["hyper"], d["valued"]) == (nrows, ncols, len(cols), hyper, valued)
        assert d["p"].tolist() == p.tolist() and d["i"].tolist() == cols.tolist()
        if hyper:
            assert d["h"].tolist() == h.tolist()
        if valued:
            assert d["x"].tolist() == x.tolist()

    check()


def test_fuse_anonymous_traverse_random_plans_match_the_oracle():
    """Property test (hypothesis): random linear plans — chains of CondTraverse hops with random flags, labels,
    aliases, interleaved Filter nodes — give the same plan from the C++ mirror and the Python restatement, the pass
    is idempotent, and every hop of the input survives in order in exactly one operator."""
    from hypothesis import given, settings, strategies as st
    from falkordb_amd import host
    from oracle import model

    hop = st.fixed_dictionaries({
        "named_edge": st.booleans(), "emit": st.booleans(), "transposed": st.sampled_from([False, False, False, True]),
        "optional": st.sampled_from([False, False, False, True]), "bidir": st.sampled_from([False, False, True]),
        "var_len": st.sampled_from([False, False, False, True]), "attrs": st.sampled_from([False, False, True]),
        "mid_named": st.sampled_from([False, False, True]), "mid_label": st.sampled_from([False, False, True]),
        "filter_above": st.sampled_from([None, None, None, "mid", "other"]), "bind": st.sampled_from([True, True, False]),
    })

    @settings(max_examples=150, deadline=None)
    @given(st.lists(hop, min_size=1, max_size=6))
    def check(hops):
        k = len(hops)
        # node aliases: n0 = a ... nk = end; intermediates anonymous unless mid_named
        nodes = []
        for i in range(k + 1):
            inner = 0 < i < k
            named = (not inner) or hops[i]["mid_named"]
            nodes.append({"alias": (f"n{i}" if named else f"_anon_n{i}"),
                          "labels": (["L"] if (inner and hops[i]["mid_label"]) else []), "attrs": False})
        ops, nid = [{"id": 0, "parent": -1, "kind": "X", "name": "Project", "refs": [nodes[k]["alias"]]}], 1
        parent = 0
        for i in range(k - 1, -1, -1):                  # the last hop sits highest in the plan
            h = hops[i]
            if h["filter_above"]:
                ref = nodes[i + 1]["alias"] if h["filter_above"] == "mid" else "zzz"
                ops.append({"id": nid, "parent": parent, "kind": "X", "name": "Filter", "refs": [ref]})
                parent, nid = nid, nid + 1
            rel = {"alias": (f"e{i}" if h["named_edge"] else f"_anon_e{i}"), "from": dict(nodes[i]), "to": dict(nodes[i + 1]),
                   "types": ["T"], "bidirectional": h["bidir"], "var_len": h["var_len"], "attrs": h["attrs"]}
            ops.append({"id": nid, "parent": parent, "kind": "CT", "rel": rel, "emit": h["emit"], "transposed": h["transposed"],
                        "optional": h["optional"], "bind": h["bind"], "siblings": [], "chain": []})
            parent, nid = nid, nid + 1
        ops.append({"id": nid, "parent": parent, "kind": "X", "name": "AllNodeScan", "refs": []})
        got, _ = host.plan_fuse(ops)
        want = model.fuse_anonymous_traverse(ops)
        assert _canon(got) == _canon(want)
        again, _ = host.plan_fuse(got)
        assert _canon(again) == _canon(got)                                   # idempotent
        # every input hop appears exactly once, in traversal order, across the surviving operators (leaf to root)
        by_id = {o["id"]: o for o in got}
        order, cur = [], [o for o in got if o["parent"] == -1][0]["id"]
        kids = {o["id"]: [c["id"] for c in got if c["parent"] == o["id"]] for o in got}
        chain_top_down = []
        while True:
            o = by_id[cur]
            if o["kind"] == "CT":
                chain_top_down.append([o["rel"]["alias"]] + [r["alias"] for r in o["chain"]])
            if not kids[cur]:
                break
            cur = kids[cur][0]
        flat = [a for grp in reversed(chain_top_down) for a in grp]
        assert flat == [(f"e{i}" if hops[i]["named_edge"] else f"_anon_e{i}") for i in range(k)]

    check()
