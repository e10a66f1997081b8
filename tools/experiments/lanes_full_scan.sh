#!/bin/bash
# the headline step (whole :P scan at RMAT-22) on 2 / 3 lanes, alternating, same box
for rep in 1 2; do for l in 2 3; do
  timeout 400 python bench.py --quick --scale 22 --steps 6 --warmup 1 --no-cpu-baseline --no-parity --no-pmc --no-lanes-sweep --opt expand_scan_lanes=$l 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes=$l ms_per_step', d['ms_per_step'], 'TTEPS', round(d['value']/1e12,3))"
done; done
