bash tools/bfs_quick.sh r03z
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_threads.py tests/test_gpu_host.py -x -q -k "dist or thread or bfs or algo" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -2
python tools/chain_bfs.py 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-pmc --no-khop --no-scale-base --no-varlen --no-roofline --steps 32 --warmup 8 --scale 26"
for m in "" "--force-dist"; do $B $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scale26', '$m', d['ms_per_step'], round(d['value']/1e9,1))"; done
