"""Golden (nnz, checksum, flops, per-hop nnz) of the FULL 1024-row batch 0 of the RMAT-26 3-hop chain bench.py times, clean and
dirty — the CPU oracle's delta_lmxm chain (oracle.expand_summary_omp over oracle/oracle_omp.c), run ONCE and committed as
tests/golden/khop26_batch0.json: a minute of 16-thread CPU time that the test suite and bench.py then do not repeat
(the largest configuration was only ever checked on 128 of its 1024 rows).

Needs a GPU box only for the INPUTS: the graph and its delta layers are the device-generated ones of bench.py khop_inputs /
tests/conftest.py (fgpu_mat_rmat is held equal to oracle.rmat_csr entry for entry at smaller scales by smoke() and
tests/test_gpu_matrix.py; the oracle's own numpy generator needs ~50 min for scale 26).  Everything that is CHECKED comes from
the oracle.   usage: python tests/golden/make_khop26_golden.py [scale]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
from conftest import _BenchGraphs  # noqa: E402
from falkordb_amd import engine  # noqa: E402
from test_gpu_scale import p_sources  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ctx = engine.Context(0)
g = _BenchGraphs(ctx)
A, _, a = g(scale)
dp, dm, hdp, hdm = g.khop_layers(scale)
src = p_sources(A.nrows, 1024)
out = {"scale": scale, "vertices": int(a.nrows), "edges": int(a.nnz), "nnz_dp": int(hdp.nnz), "nnz_dm": int(hdm.nnz), "rows": len(src), "hops": 3,
       "graph": "fgpu_mat_rmat(scale, 16, 0x5EED1234 + scale); dm = fgpu_mat_sample(0xD3170 + scale, 1000); dp = rng(0xADD5 + scale) "
                "coordinates outside A (bench.py khop_inputs)",
       "sources": "the first 1024 ids with mix64(id) % 16 == 0",
       "sources_sha256": hashlib.sha256(np.ascontiguousarray(src).tobytes()).hexdigest(),
       "colidx_sha256": hashlib.sha256(np.ascontiguousarray(a.colidx).tobytes()).hexdigest(),
       "oracle": "oracle.expand_summary_omp (oracle/oracle_omp.c orc_mxm_omp / orc_merge_omp, 64 source rows per chunk)"}
for name, layers in (("clean", [(a, None, None)] * 3), ("dirty", [(a, hdp, hdm)] * 3)):
    t = time.time()
    nnz, cs, flops, hop_nnz = oracle.expand_summary_omp(src, layers)
    out[name] = {"nnz": int(nnz), "checksum": int(cs), "flops": int(flops), "hop_nnz": [int(x) for x in hop_nnz],
                 "oracle_seconds": round(time.time() - t, 1), "threads": oracle.omp_threads()}
    print(name, out[name], flush=True)
path = os.path.join(ROOT, "tests", "golden", "khop%d_batch0.json" % scale)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for p in (path, os.path.join(ROOT, "gpurun_out", "khop%d_batch0.json" % scale)):
    with open(p, "w") as f:
        json.dump(out, f, indent=1)
print("wrote", path)
