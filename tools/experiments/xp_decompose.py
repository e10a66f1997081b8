import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from falkordb_amd import engine
ctx = engine.Context(0)
A = ctx.mat_rmat(22, 16, 0x5EED1234 + 22)
n = A.nrows
srcs = bench.p_label_sources(n)
batches = [srcs[j * 1024:(j + 1) * 1024] for j in range(8)]
def run(tag, **opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    for b in batches[:2]: engine.expand_count(ctx, b, [A] * 3)
    ctx.prof_enable(True)
    for b in batches: engine.expand_count(ctx, b, [A] * 3)
    prof = ctx.prof_read(); ctx.prof_enable(False)
    rows = {k["kernel"]: round(k["ms"] / k["launches"] * 1e3, 1) for k in prof if k["kernel"].startswith(("xp_", "bp_pull_kernel<dense"))}
    print(tag, rows, flush=True)
run("default")
run("no LDS flush", expand_xcd_dbg=1)
run("gathers from 256 hot rows", expand_xcd_dbg=2)
run("gathers from 64K rows (4 MB)", expand_xcd_dbg=4)
run("hot rows + no flush", expand_xcd_dbg=3)
for wg in (1, 2, 3, 4, 5, 6):
    run("workgroups per CU %d" % wg, expand_xcd_dbg=wg << 8)
run("plain pull", expand_xcd_dbg=0, expand_xcd=0)
