# pass-width sweep of the whole-frontier call, one lane, per-kernel table: what hop 2 / the count hop cost per pass at each width
mkdir -p gpurun_out
timeout 600 python tools/scan_sweep.py --scale 22 --nsrc 65536 --lanes 1 --rows 256,512,1024,2048,4096 --prof --reps 1 --out gpurun_out/w1_22.json > gpurun_out/w1_22.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/w1_22.json'))
for r in d['runs']:
    print(r['rows'], 'passes', r['passes'], 'ms', r['ms'], 'TTEPS %.2f' % (r['TEPS']/1e12), r['ok'])
    for k in r.get('kernels', []):
        print('    %-40s %9.3f ms %5d x %9.1f us' % (k['kernel'], k['ms'], k['launches'], k['us_per_launch']))
PY
