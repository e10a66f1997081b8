"""Per-BFS kernel sequences from a rocprofv3 kernel trace: F = fused level kernel (P / L = the direction-named push / pull launches of the profiled pass), c = ctrl, S/M = step/commit."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seqs = []; cur = None; last_end = None
for r in rows:
    n = r['Kernel_Name']
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (en - st) / 1000
    if 'bfs_init' in n or 'fused_begin' in n:
        if cur: seqs.append(cur)
        cur = [('B', round(d, 1), round((st - last_end) / 1000, 1) if last_end else 0)]; last_end = en
    elif cur is not None and ('fused' in n or 'ctrl' in n or 'step' in n or 'commit' in n or 'tiny' in n):
        gap = (st - last_end) / 1000 if last_end else 0
        cur.append(('T' if 'tiny' in n else 'P' if ('fused' in n and ', 1>' in n) else 'L' if ('fused' in n and ', 2>' in n) else 'F' if 'fused' in n else 'c' if 'ctrl' in n else 'S' if 'step' in n else 'M', round(d, 1), round(gap, 1)))
        last_end = en
seqs.append(cur)
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for s in seqs[lo:lo + 6]:
    print(' '.join(f"{k}{d}(+{g})" for k, d, g in s), ' total', round(sum(d + g for _, d, g in s), 1))
