mkdir -p gpurun_out
for w in 6; do
  for s in 22 26; do
    FGPU_OPTS=bfs_wgs_per_cu=$w timeout 600 python bench.py --leg bfs --no-roofline --scale $s --steps 64 --warmup 16 2>/dev/null | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('wgs',$w,'scale',$s,'ms',l.get('ms_per_step'),'GTEPS',round(l.get('value',0)/1e9,1))"
  done
done
timeout 900 python -m pytest tests/test_gpu_traversal.py -x -q -m gpu -k "bfs" 2>&1 | tail -2
