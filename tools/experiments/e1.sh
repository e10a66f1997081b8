mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_traversal.py -x -q -m gpu -k "emission_forms or stream_chunks" 2>&1 | tail -3
timeout 300 python tools/emit_ab.py 22 3 1024 3 2>&1 | tail -3 | tee gpurun_out/e1_22_3.log
timeout 300 python tools/emit_ab.py 24 2 1024 6 2>&1 | tail -3 | tee gpurun_out/e1_24_2.log
timeout 300 python tools/emit_ab.py 22 2 1024 6 2>&1 | tail -3 | tee gpurun_out/e1_22_2.log
