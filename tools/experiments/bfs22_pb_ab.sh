#!/bin/bash
# RMAT-22 BFS, 64 roots: bfs_pb 0 / 1 (default: no launches below 2^24 vertices) / 2 (listing + blocked push forced on), two runs each.
for o in ${MODES:-1 2 1 2}; do
  timeout 300 python bench.py --leg bfs --scale 22 --steps 128 --warmup 16 --no-roofline --opt bfs_pb=$o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bfs_pb=$o RMAT-22 ms_per_step', d['ms_per_step'], 'GTEPS', round(d['value']/1e9,1))"
done
