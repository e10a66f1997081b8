timeout 900 python -m pytest tests/test_gpu_matrix.py tests/test_gpu_host.py -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
python tools/bench_paths.py merge 22 2>/dev/null | cut -c1-250
