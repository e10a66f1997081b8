B="python bench.py --no-cpu-baseline --no-pmc --no-khop --no-scale-base --no-varlen --no-roofline --steps 32 --warmup 8"
for sc in 24 26; do $B --scale $sc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scale$sc', d['ms_per_step'], round(d['value']/1e9,1))"; done
$B --scale 26 --force-dist 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scale26 dist', d['ms_per_step'], round(d['value']/1e9,1))"
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_threads.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
