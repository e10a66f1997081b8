cd /root/repo
timeout 300 python -m pytest tests/test_gpu_matrix.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "partitioned_by_xcd" 2>&1 | tail -2
timeout 300 python tools/time_transpose.py 22 2>&1 | grep -v "coo+sort" | sed -n 5,12p
