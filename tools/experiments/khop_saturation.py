#!/usr/bin/env python3
"""Driver of khop_saturation.c: the bench's RMAT graph (scale argv[1], default 22), the first 1024 :P sources, two hops of the
oracle's chain -> the bit state X after hop 2 -> dumps for the C counter.  Test infrastructure only (uses oracle/)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/w"
t0 = time.time()
a = oracle.rmat_csr(scale, 16, 0x5EED1234 + scale)
n = a.nrows
ids = np.arange(n, dtype=np.uint64)
src = ids[oracle.mix64(ids) % np.uint64(16) == 0][:1024]
f, flops, hop = oracle.expand_omp(src, [(a, None, None)] * 2)
print(f"scale {scale} n {n} nnz {a.nnz}; 2-hop state {f.nnz} entries ({time.time()-t0:.0f} s)", flush=True)
W = 16
x = np.zeros((n, W), dtype=np.uint64)
rows = np.repeat(np.arange(1024, dtype=np.uint64), np.diff(f.rowptr).astype(np.int64))
cols = f.colidx.astype(np.int64)
np.bitwise_or.at(x, (cols, (rows >> np.uint64(6)).astype(np.int64)), np.uint64(1) << (rows & np.uint64(63)))
at = oracle.transpose(a)
np.asarray(at.rowptr, dtype=np.uint64).tofile(f"{out}/sat_rp.bin")
np.asarray(at.colidx, dtype=np.uint64).tofile(f"{out}/sat_ci.bin")
x.tofile(f"{out}/sat_x.bin")
exe = f"{out}/khop_saturation"
subprocess.run(["gcc", "-O3", "-fopenmp", "-o", exe, os.path.join(os.path.dirname(os.path.abspath(__file__)), "khop_saturation.c")], check=True)
print(subprocess.run([exe, f"{out}/sat_rp.bin", f"{out}/sat_ci.bin", f"{out}/sat_x.bin", "1024"], capture_output=True, text=True).stdout)
