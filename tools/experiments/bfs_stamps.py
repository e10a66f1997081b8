"""Per-workgroup phase stamps of the fused BFS level kernel (library built with -DFGPU_BFS_STAMPS: compile bfs.hip with\nthe flag into falkordb_amd/lib/obj/bfs.o and relink; see the comment at DBG_STAMP in bfs.hip)."""
import sys, ctypes as C, collections
sys.path.insert(0, ".")
import numpy as np, torch
from falkordb_amd import engine
from bench import pick_roots
ctx = engine.Context(0)
A = ctx.mat_rmat(22); At = A.transpose()
roots = pick_roots(A, 64)
plan = engine.BfsPlan(ctx, A, At)
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
lib = ctx.lib
lib.fgpu_debug_bfs_stamps.argtypes = [C.c_void_p]
for ri, L in [(0, 3), (0, 4), (17, 3), (17, 4), (17, 5)]:
    root = roots[ri]
    plan.run(root)
    buf.zero_()
    lib.fgpu_debug_bfs_stamps(C.c_void_p(buf.data_ptr()))
    plan.run(root, L)
    lib.fgpu_debug_bfs_stamps(None)
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(-1, 8)
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min()
    st = (b[:, 0] - t0) / 100.0
    en = (b[:, 5] - t0) / 100.0
    hw = b[:, 7] & 0xFFFFFFFF
    xcc = (b[:, 7] >> 32) & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7   # gfx9 HW_ID layout: cu_id[11:8] sh_id[12] se_id[15:13]
    print(f"root#{ri} level {L}: {len(b)} wgs; start percentiles", [round(float(np.percentile(st, q)), 1) for q in (50, 75, 80, 85, 90, 95, 100)], "late(>5us):", int((st > 5).sum()))
    key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    cnt = collections.Counter(key)
    early = collections.Counter(k for k, s_ in zip(key, st) if s_ <= 5)
    print("  distinct CUs seen:", len(cnt), " WGs per CU histogram:", sorted(collections.Counter(cnt.values()).items()), " early WGs per CU histogram:", sorted(collections.Counter(early.values()).items()))
    perx = collections.Counter(xcc.tolist()); print("  WGs per XCC:", sorted(perx.items()))
    latex = collections.Counter(x for x, s_ in zip(xcc.tolist(), st) if s_ > 5); print("  late WGs per XCC:", sorted(latex.items()))
    # blockIdx of late WGs
    idx = np.nonzero(buf.cpu().numpy().reshape(-1, 8)[:, 0] > 0)[0]
    late_idx = idx[st > 5]
    print("  late blockIdx range:", late_idx.min() if len(late_idx) else None, late_idx.max() if len(late_idx) else None, " first few:", late_idx[:10].tolist())
    print("  end time percentiles:", [round(float(np.percentile(en, q)), 1) for q in (10, 50, 90, 100)])
    for k, name in [(1, "push items done"), (2, "push hubs done"), (3, "level work done"), (4, "ticket done"), (5, "end")]:
        x = (b[:, k] - t0) / 100.0
        x = x[(b[:, k] > 0) & (x > 0)]
        if len(x): print(f"   {name:16s} p10 {np.percentile(x,10):6.1f} p50 {np.median(x):6.1f} p90 {np.percentile(x,90):6.1f} p99 {np.percentile(x,99):6.1f} max {x.max():6.1f}  n={len(x)}")
    w = (b[:, 3] - t0) / 100.0
    order = np.argsort(-w)[:8]
    print("   slowest workgroups (blockIdx, level-work-done us):", [(int(idx[i]), round(float(w[i]), 1)) for i in order])
