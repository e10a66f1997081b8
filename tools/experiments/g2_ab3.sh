mkdir -p gpurun_out
run() { # name, opts...
  name=$1; shift
  args=""
  for o in "$@"; do args="$args --opt $o"; done
  timeout 300 python tools/scan_sweep.py --scale ${SCALE:-22} --nsrc ${NSRC:-16384} --lanes 1 --prof $args --out gpurun_out/h3_$name.json > gpurun_out/h3_$name.log 2>&1
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/h3_$name.json'))
    r=d['runs'][0]
    k={x['kernel']:x['us_per_launch'] for x in r.get('kernels',[])}
    print('$name', 'ms',r['ms'],'TEPS %.2fT'%(r['TEPS']/1e12), 'sparse',k.get('bp_pull_kernel<sparse>'),'groups',k.get('sparse pull: row groups'),'stream',k.get('xp_stream_kernel'),'ok',r['ok'])
except Exception as e:
    print('$name', 'failed', e); print(open('gpurun_out/h3_$name.log').read()[-600:])
PY
}
run base
run g2_4_items_old expand_groups2=4 expand_items2=0
run g2_4_items_new expand_groups2=4 expand_items2=1
run g2_8_items_new expand_groups2=8 expand_items2=1
