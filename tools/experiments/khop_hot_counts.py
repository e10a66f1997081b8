#!/usr/bin/env python3
"""CPU counts behind DESIGN.md §4.3's hop-3 experiments: on the bench's own RMAT graph and first 1024-row
:P batch, (a) how many of A''s entries sit in rows whose FINAL hop-3 row Y[v] is all-ones, and after how many gathers (in
storage order) such a row saturates; (b) which share of the entries has BOTH endpoints in the H hottest vertices (hot by
out-degree for the gathered side u, by in-degree for the row side v).  Test infrastructure only (uses oracle/)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
t0 = time.time()
a = oracle.rmat_csr(scale, 16, 0x5EED1234 + scale)
n = a.nrows
rp = np.asarray(a.rowptr, dtype=np.int64); ci = np.asarray(a.colidx, dtype=np.int64)
nnz = len(ci)
outdeg = np.diff(rp)
indeg = np.bincount(ci, minlength=n)
print(f"scale {scale}: n {n} nnz {nnz} build {time.time()-t0:.1f}s", flush=True)
rows = np.repeat(np.arange(n, dtype=np.int64), outdeg)      # edge (u=rows -> v=ci); A' entry (v, u)
order_out = np.argsort(-outdeg, kind="stable"); rank_out = np.empty(n, np.int64); rank_out[order_out] = np.arange(n)
order_in = np.argsort(-indeg, kind="stable"); rank_in = np.empty(n, np.int64); rank_in[order_in] = np.arange(n)
ru = rank_out[rows]; rv = rank_in[ci]
print("H_u (hot gathered rows) x H_v (hot destination rows): share of entries with u hot | v hot | both hot")
for H in (16384, 32768, 65536, 131072, 262144, 524288):
    hu = ru < H
    for Hv in (H, 4 * H):
        hv = rv < Hv
        print(f"  H_u {H:7d} H_v {Hv:8d}: u-hot {hu.mean():.3f}  v-hot {hv.mean():.3f}  both {np.mean(hu & hv):.3f}", flush=True)
