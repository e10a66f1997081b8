"""One BFS root under rocprofv3 --kernel-trace: the kernels of the LAST of a few searches, in order, with durations.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/experiments/pb_one.py [scale] [root_index] [reps]
then:            python tools/experiments/pb_one.py --read DIR/t_kernel_trace.csv"""
import csv
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last search: from the last bfs_fused_begin_kernel on
    last = max(i for i, r in enumerate(rows) if "bfs_fused_begin" in r["Kernel_Name"])
    t0 = int(rows[last]["Start_Timestamp"])
    for r in rows[last:]:
        n = r["Kernel_Name"].split("(")[0].replace("void fgpu::", "").replace("fgpu::", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  {n}  grid {r.get('Grid_Size', '')}")
    sys.exit(0)
sys.path.insert(0, ".")
from falkordb_amd import engine
from bench import pick_roots

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ri = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = engine.Context(0)
A = ctx.mat_rmat(scale)
At = A.transpose()
root = pick_roots(A, 64)[ri]
plan = engine.BfsPlan(ctx, A, At)
for _ in range(reps):
    plan.run(root)
print(root, plan.stats(), ctx.get_option("bfs_pb_last_levels"))
