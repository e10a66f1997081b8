"""Every kernel of a few BFS searches from a rocprofv3 kernel trace (name initial, duration us, gap us): B begin, F fused level,
l list, p listed-candidate pull, x/c/s/a blocked-push prefix / count / scatter / apply, T tiny.  usage: bfs_trace_all.py trace.csv [first] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def tag(n):
    for k, t in (('fused_begin', 'B'), ('bfs_init', 'B'), ('fused', 'F'), ('pb_list', 'l'), ('bfs_lp', 'p'), ('pb_prefix', 'x'), ('pb_count', 'c'),
                 ('pb_scatter', 's'), ('pb_apply', 'a'), ('tiny', 'T')):
        if k in n: return t
    return None
seqs, cur, last = [], None, None
for r in rows:
    t = tag(r['Kernel_Name'])
    if t is None: continue
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t == 'B':
        if cur: seqs.append(cur)
        cur = []
    if cur is None: continue
    cur.append((t, (en - st) / 1000, (st - last) / 1000 if last and cur else 0.0))
    last = en
if cur: seqs.append(cur)
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for s in seqs[lo:lo + n]:
    print(' '.join(f"{k}{d:.1f}(+{g:.1f})" for k, d, g in s), ' | kernels', round(sum(d for _, d, _ in s), 1), 'span', round(sum(d + g for _, d, g in s[1:]) + s[0][1], 1))
