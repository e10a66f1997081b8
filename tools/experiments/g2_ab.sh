set -x
mkdir -p gpurun_out
FGPU_OPTS=expand_groups2=4 timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_traversal.py -x -q -m gpu > gpurun_out/g2_pytest.log 2>&1
tail -3 gpurun_out/g2_pytest.log
for g in 0 4 8; do
  timeout 300 python tools/scan_sweep.py --scale 22 --nsrc 32768 --lanes 1,3 --prof --opt expand_groups2=$g --out gpurun_out/g2_22_$g.json > gpurun_out/g2_22_$g.log 2>&1
  tail -2 gpurun_out/g2_22_$g.log | cut -c1-300
done
for g in 0 4; do
  timeout 400 python tools/scan_sweep.py --scale 26 --nsrc 4096 --lanes 1 --prof --opt expand_groups2=$g --out gpurun_out/g2_26_$g.json > gpurun_out/g2_26_$g.log 2>&1
  tail -1 gpurun_out/g2_26_$g.log | cut -c1-300
done
