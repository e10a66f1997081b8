mkdir -p gpurun_out
bash tools/experiments/h4.sh
SCALE=26 ; timeout 400 python tools/scan_sweep.py --scale 26 --nsrc 4096 --lanes 1 --prof --out gpurun_out/h5_26.json > gpurun_out/h5_26.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/h5_26.json'))
for r in d['runs']:
    k={x['kernel']:x['us_per_launch'] for x in r.get('kernels',[])}
    print('RMAT-26 lanes',r['lanes'],'ms',r['ms'],'TEPS %.2fT'%(r['TEPS']/1e12), 'sparse',k.get('bp_pull_kernel<sparse>'),'groups',k.get('sparse pull: row groups'),'stream',k.get('xp_stream_kernel'),'fold',k.get('xp_fold_kernel'),'ok',r['ok'])
PY
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_traversal.py -x -q -m gpu 2>&1 | tail -3
