#!/bin/bash
# BFS 64-root loop, baseline library (.ab/libfgpu_base.so through FGPU_LIB) against the built one, same box, alternating.
# usage: bash tools/experiments/bfs_lib_ab.sh "22 24 26"
for sc in ${1:-22}; do st=128; [ $sc = 26 ] && st=64
  for rep in 1 2; do for which in base new; do
    L=""; [ $which = base ] && L=$GRAFT_REPO_ROOT/.ab/libfgpu_base.so
    FGPU_LIB=$L timeout 400 python bench.py --leg bfs --scale $sc --steps $st --warmup 16 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which RMAT-$sc ms_per_step', d['ms_per_step'], 'GTEPS', round(d['value']/1e9,1))"
  done; done
done
