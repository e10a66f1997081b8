set -x
mkdir -p gpurun_out
for nt in 0 1 2 3 4 7; do
  timeout 300 python tools/scan_sweep.py --scale 22 --nsrc 32768 --lanes 1,3 --prof --opt expand_nt=$nt --out gpurun_out/nt22_$nt.json > gpurun_out/nt22_$nt.log 2>&1
done
for nt in 0 3 7; do
  timeout 400 python tools/scan_sweep.py --scale 26 --nsrc 4096 --lanes 1,3 --prof --opt expand_nt=$nt --out gpurun_out/nt26_$nt.json > gpurun_out/nt26_$nt.log 2>&1
done
