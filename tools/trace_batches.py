"""One k-hop batch on the GPU timeline, from a rocprofv3 kernel trace (kernel_trace.csv): the kernels of a batch in launch order
with their duration and the idle gap before each (us).  A batch starts at build_frontier / the first kernel after a gap > 150 us.
usage: python tools/trace_batches.py <kernel_trace.csv> [first_batch] [count]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
batches, cur, last = [], [], None
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - last) / 1000 if last else 0
    name = re.sub(r'\(.*', '', r['Kernel_Name'])
    name = re.sub(r'^void (fgpu::)?', '', name)
    if gap > 150 and cur:
        batches.append(cur); cur = []
    cur.append((name[:44], (en - st) / 1000, gap))
    last = en
batches.append(cur)
big = [b for b in batches if any('bp_pull_kernel' in k[0] or 'xp_stream' in k[0] for k in b)]
print(len(batches), 'segments,', len(big), 'with a bit-parallel pull')
for b in big[first:first + count]:
    busy = sum(d for _, d, _ in b); idle = sum(g for _, _, g in b[1:])
    print(f'--- batch: {len(b)} kernels, busy {busy:.0f} us, idle between kernels {idle:.0f} us, lead-in gap {b[0][2]:.0f} us')
    for k, d, g in b:
        if d > 0 or g > 8: print(f'   {k:44s} {d:8.1f} us   (+{g:.1f} idle before)')
