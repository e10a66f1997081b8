"""Sweep of the counting transpose's digit split (option transpose_wb = low-digit bits) on RMAT-22: wall ms and the
per-kernel HIP-event times.  usage: python tools/transpose_wb_sweep.py"""
import sys,time
sys.path.insert(0,'.')
from falkordb_amd import engine
ctx=engine.Context(0)
A=ctx.mat_rmat(22); ctx.sync()
for wb in (0, 9, 10, 11, 12, 13):
    ctx.set_option("transpose_wb", wb)
    ts=[]
    for _ in range(5):
        t0=time.perf_counter(); T=A.transpose(); ctx.sync(); ts.append(time.perf_counter()-t0); T.free()
    ctx.prof_enable(True); T=A.transpose(); pr={k["kernel"]:round(k["ms"],3) for k in ctx.prof_read()}; ctx.prof_enable(False); T.free()
    print(wb, round(min(ts)*1e3,3), pr, flush=True)
