#!/usr/bin/env python3
"""HBM bytes per launch of the k-hop kernels at a given scale, by the bench's own live-PMC procedure (two rocprofv3 --pmc passes
over `bench.py --pmc-child`): usage  python tools/pmc_khop.py <scale>  ->  one JSON object on stdout."""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
res = bench.live_pmc(types.SimpleNamespace(scale=scale, no_bfs=True), timeout_s=400)
print(json.dumps({"scale": scale, "procedure": "bench.live_pmc: FETCH_SIZE x2 (gfx950) + WRITE_SIZE, bytes per launch", "kernels": res}))
