timeout 900 python -m pytest tests/test_gpu_pagerank.py tests/test_gpu_host.py tests/test_gpu_matrix.py -x -q -k "pagerank or merge or flush" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
python tools/bench_paths.py pagerank 22 2>/dev/null | cut -c1-400
