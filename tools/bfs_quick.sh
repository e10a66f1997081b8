#!/bin/bash
# BFS-only loop for kernel work (run on the GPU box): the BFS parity tests, then the blind 64-root loop under a kernel trace.
# usage: bash tools/bfs_quick.sh <tag> [--skip-tests] [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
if [ "$1" == "--skip-tests" ]; then shift; else
  timeout 600 python -m pytest tests/test_gpu_traversal.py -x -q -k "bfs" > $out/pytest.log 2>&1; grep -E "passed|failed|error" $out/pytest.log | tail -2
fi
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --leg bfs --no-roofline --steps 128 --warmup 16"
for i in 1 2; do $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'GTEPS', round(d['value']/1e9,1))"; done
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- $B "$@" > $out/b.json 2> $out/b.err
python tools/trace_levels.py $out/trace/t_kernel_trace.csv 100 | head -6
rm -rf $out/trace
