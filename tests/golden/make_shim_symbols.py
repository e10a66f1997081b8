#!/usr/bin/env python3
"""Which link-time symbols does the reference's GraphBLAS wrapper need?  Parses the `use super::{...}` / `use
crate::graph::graphblas::{...}` import lists of graph/src/graph/graphblas/{matrix,vector,tensor,versioned_matrix}.rs (and the
names their #[cfg(test)] modules pull from `super::super`) and keeps the names that bindgen declares as an `extern "C"` function
or static in graphblas/mod.rs — types, enums and constants need no symbol.  Run in the build container (the reference tree
is not on the GPU box); writes tests/golden/shim_symbols.json, which tests/test_gpu_shim.py holds against `nm -D` of the
tier-2 library.  Usage: python tests/golden/make_shim_symbols.py [/root/reference]"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
gb = os.path.join(ref, "graph", "src", "graph", "graphblas")
mod = open(os.path.join(gb, "mod.rs")).read()
functions = set(re.findall(r"pub fn ((?:GrB|GxB|GB)_\w+)\s*\(", mod))
statics = set(re.findall(r"pub static (?:mut )?((?:GrB|GxB)_\w+)\s*:", mod))
out = {"_source": "graph/src/graph/graphblas/{matrix,vector,tensor,versioned_matrix}.rs import lists x mod.rs extern declarations",
       "files": {}}
allf, alls = set(), set()
for fn in ("matrix.rs", "vector.rs", "tensor.rs", "versioned_matrix.rs"):
    src = open(os.path.join(gb, fn)).read()
    names = set()
    for block in re.findall(r"use (?:super(?:::super)?|crate::graph::graphblas)::\{(.*?)\};", src, flags=re.S):
        names |= set(re.findall(r"\b((?:GrB|GxB)_\w+)\b", block))
    for single in re.findall(r"use (?:super(?:::super)?|crate::graph::graphblas)::((?:GrB|GxB)_\w+);", src):
        names.add(single)
    f = sorted(names & functions)
    s = sorted(names & statics)
    out["files"][fn] = {"functions": f, "globals": s}
    allf |= set(f)
    alls |= set(s)
# LAGraph: the entry points matrix.rs (init / shutdown) and algo_procedures.rs call, split by the binding file that declares them
algo = open(os.path.join(ref, "graph", "src", "runtime", "functions", "algo_procedures.rs")).read()
matrix_rs = open(os.path.join(gb, "matrix.rs")).read()
for key, fn in (("lagraph", "lagraph_bindings.rs"), ("lagraphx", "lagraphx_bindings.rs")):
    declared = set(re.findall(r"pub fn (LAGr\w*_\w+)\s*\(", open(os.path.join(gb, fn)).read()))
    used = set(re.findall(key + r"_bindings::(LAGr\w*_\w+)", algo))
    if key == "lagraph":
        used |= set(re.findall(r"\b(LAGraph_\w+)\b", matrix_rs))
    out[key] = sorted(used & declared)
# the GraphBLAS calls inside the algo.BFS (:1017-1165) and algo.pageRank (:687-783) procedures and their helpers (:385-447)
lines = algo.split("\n")
span = "\n".join(lines[384:447] + lines[686:783] + lines[1016:1165])
out["algo_bfs_pagerank_graphblas"] = sorted(set(re.findall(r"\b((?:GrB|GxB)_\w+)\b", span)) & functions)
out["functions"] = sorted(allf)
out["globals"] = sorted(alls)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim_symbols.json")
json.dump(out, open(dst, "w"), indent=1)
print(f"{len(allf)} functions, {len(alls)} globals, {len(out['lagraph'])} + {len(out['lagraphx'])} LAGraph entry points -> {dst}")
