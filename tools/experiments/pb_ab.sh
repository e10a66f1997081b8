# BFS A/B on one box: parity tests, then the 64-root loop at RMAT-26 / 24 / 22 (usage: bash tools/experiments/pb_ab.sh)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_threads.py tests/test_gpu_host.py -x -q -m gpu -k "bfs" 2>&1 | tail -2
for sc in 26 24 22; do st=64; [ $sc = 22 ] && st=128; for i in 1 2; do timeout 400 python bench.py --leg bfs --scale $sc --steps $st --warmup 8 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RMAT-$sc ms_per_step', d['ms_per_step'], 'GTEPS', round(d['value']/1e9,1))"; done; done | tee gpurun_out/bfs_ab.log
bash tools/experiments/pb_one.sh 2>&1 | tail -22
