mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python tools/scan_sweep.py --scale 22 --nsrc 16384 --lanes 1,3 --prof --out gpurun_out/h4_$i.json > gpurun_out/h4_$i.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/h4_$i.json'))
for r in d['runs']:
    k={x['kernel']:x['us_per_launch'] for x in r.get('kernels',[])}
    print('lanes',r['lanes'],'ms',r['ms'],'TEPS %.2fT'%(r['TEPS']/1e12), 'sparse',k.get('bp_pull_kernel<sparse>'),'groups',k.get('sparse pull: row groups'),'stream',k.get('xp_stream_kernel'),'ok',r['ok'])
PY
done
