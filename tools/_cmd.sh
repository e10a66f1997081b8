timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_threads.py tests/test_gpu_host.py tests/test_gpu_shim.py -x -q -k "bfs or thread or algo or shim" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
python bench.py --no-cpu-baseline --no-pmc --no-khop --no-scale-base --no-varlen --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['bfs_host_arrays'])[:300])"
