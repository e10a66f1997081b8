timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_host.py tests/test_gpu_threads.py -x -q -k "expand or khop or varlen or levels or thread or reach" > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -3
python bench.py --no-cpu-baseline --no-pmc --no-scale-base --no-roofline --steps 8 --warmup 2 --khop-batches 8 --khop-extra-scales "" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
m=d['khop_materialised']
for k,v in m.items():
    if isinstance(v,dict) and 'ms_per_batch' in v: print(k, v['ms_per_batch'], [(x['kernel'],x['avg_launch_us']) for x in v.get('kernels',[])][:6])
print('c5', d['config5_varlen']['ms_per_batch_per_rank'], d['config5_varlen']['TEPS']/1e9)
print('khop24', d['khop_match']['clean']['ms_per_batch'], d['khop_match']['dirty']['ms_per_batch'])
"
