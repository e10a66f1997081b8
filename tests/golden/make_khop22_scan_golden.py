"""Golden (nnz, checksum, flops) of the WHOLE :P scan of the RMAT-22 3-hop chain — every source with mix64(id) % 16 == 0
(261 6xx rows) in ONE call, the workload bench.py times since round 6 (SURVEY.md §7 hard part 1 / §8d: "all active sources of a
scan") — from the CPU oracle's delta_lmxm chain (oracle/oracle_omp.c orc_mxm_omp, 64 source rows per chunk, every row hashed
with its row index in the whole call), run ONCE and committed as tests/golden/khop22_scan.json: ~15 minutes of 16-thread CPU
time that neither the test suite nor bench.py repeats.  The sums are also recorded per slab of 1024 consecutive source rows
(they add up: nnz and flops plainly, the checksum mod 2^64), so a mismatch can be localised and a prefix of the scan can be
checked on its own.

Needs a GPU box only for the INPUTS (the device-generated graph of bench.py khop_inputs / tests/conftest.py, held equal to
oracle.rmat_csr entry for entry at smaller scales by smoke() and tests/test_gpu_matrix.py; pinned here by sha256 of its
column ids).  Everything that is CHECKED comes from the oracle.   usage: python tests/golden/make_khop22_scan_golden.py [scale]"""
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

U64 = np.uint64
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # (a prefix of the scan, for a dry run)
ctx = engine.Context(0)
g = _BenchGraphs(ctx)
A, _, a = g(scale)
ids = np.arange(a.nrows, dtype=U64)
src = ids[oracle.mix64(ids) % U64(16) == 0]
if limit:
    src = src[:limit]
threads = 0
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    threads = 0 if q == "max" else max(1, int(float(q) / float(per)))
except (OSError, ValueError):
    pass
layers = [(a, None, None)] * 3
SLAB, CHUNK = 1024, 64
slabs = []
t0 = time.time()
tot = [0, 0, 0]
for s0 in range(0, len(src), SLAB):
    s1 = min(s0 + SLAB, len(src))
    nnz = cs = flops = 0
    for c0 in range(s0, s1, CHUNK):
        c1 = min(c0 + CHUNK, s1)
        f = oracle.build_csr(len(src), a.nrows, np.arange(c0, c1, dtype=U64), src[c0:c1])
        for (m, dp, dm) in layers:
            f, fl = oracle.delta_lmxm_omp(f, m, dp, dm, threads)
            flops += fl
        nnz += f.nnz
        cs = (cs + oracle.checksum_omp(f, threads)) & 0xFFFFFFFFFFFFFFFF
    slabs.append([int(nnz), int(cs), int(flops)])
    tot = [tot[0] + nnz, (tot[1] + cs) & 0xFFFFFFFFFFFFFFFF, tot[2] + flops]
    if (s0 // SLAB) % 16 == 0:
        print("slab", s0 // SLAB, "of", (len(src) + SLAB - 1) // SLAB, "t", round(time.time() - t0, 1), "flops so far", tot[2], flush=True)
out = {"scale": scale, "vertices": int(a.nrows), "edges": int(a.nnz), "rows": int(len(src)), "hops": 3, "layers": "clean",
       "graph": "fgpu_mat_rmat(scale, 16, 0x5EED1234 + scale) (bench.py khop_inputs)",
       "sources": "every id with mix64(id) % 16 == 0, ascending (the synthetic label :P of SURVEY.md §8d)",
       "sources_sha256": hashlib.sha256(np.ascontiguousarray(src).tobytes()).hexdigest(),
       "colidx_sha256": hashlib.sha256(np.ascontiguousarray(a.colidx).tobytes()).hexdigest(),
       "oracle": "oracle.delta_lmxm_omp chain (oracle/oracle_omp.c orc_mxm_omp), 64 source rows per chunk, rows hashed by their index "
                 "in the whole call (oracle.checksum_omp)",
       "nnz": int(tot[0]), "checksum": int(tot[1]), "flops": int(tot[2]),
       "slab_rows": SLAB, "slabs": slabs, "oracle_seconds": round(time.time() - t0, 1), "threads": threads or oracle.omp_threads()}
name = "khop%d_scan.json" % scale if not limit else "khop%d_scan_first%d.json" % (scale, limit)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for p in (os.path.join(ROOT, "tests", "golden", name), os.path.join(ROOT, "gpurun_out", name)):
    with open(p, "w") as f:
        json.dump(out, f)
print("wrote", name, {k: out[k] for k in ("rows", "nnz", "checksum", "flops", "oracle_seconds", "threads")})
# the engine, right here, on the same inputs (informational: the test suite and bench.py do the real comparison)
got = engine.expand_count(ctx, src, [A] * 3)
print("engine", got, "match", got == (out["nnz"], out["checksum"], out["flops"]))
