import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
seqs=[];cur=None
for r in rows:
    n=r['Kernel_Name']
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
    if 'bfs_init' in n:
        if cur: seqs.append(cur)
        cur=[]
    elif cur is not None and ('fused' in n or 'ctrl' in n or 'step' in n or 'commit' in n):
        cur.append(('F' if 'fused' in n else 'c' if 'ctrl' in n else 'S' if 'step' in n else 'M', round(d,1)))
seqs.append(cur)
lo=int(sys.argv[2]) if len(sys.argv)>2 else 80
for s in seqs[lo:lo+6]:
    print(' '.join(f"{k}{d}" for k,d in s))
