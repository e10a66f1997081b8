"""profiles/traffic.json from a bench line that carries live PMC results: the fallback bench.py uses when the live
rocprofv3 --pmc passes cannot run on a box — valid only for the kernel sources it was measured on (their hash is
recorded and checked by bench.py committed_traffic).

usage: python tools/make_traffic.py profiles/<tag>_bench.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src = sys.argv[1]
d = json.loads(open(src).read().strip().splitlines()[-1])
pmc = dict(d.get("pmc") or {})
khop = d.get("khop_match") or {}
scale = d["config"]["scale"]
by_dir = {x["kernel"]: x for x in d["roofline"]["by_direction"]}
out = {"_source": f"live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes recorded in {src} (bench.py live_pmc); FETCH_SIZE "
                  f"doubled per MI355X_MICROARCH.md (gfx950)",
       "_csrc_sha256": bench.csrc_hash(),
       f"rmat{scale}": {k: v for k, v in pmc.items() if isinstance(v, dict)}}
tot = sum(by_dir[k]["launches"] * pmc[k]["hbm_bytes_per_dispatch"] for k in by_dir if k in pmc)
cnt = sum(by_dir[k]["launches"] for k in by_dir if k in pmc)
if cnt:
    out[f"rmat{scale}"]["bfs_fused_kernel"] = {"hbm_bytes_per_dispatch": int(tot / cnt), "dispatches": cnt,
                                               "note": "push / pull launches weighted by that run's launch counts"}
if khop.get("pmc"):
    out[f"rmat{khop['scale']}"] = {k: v for k, v in khop["pmc"].items() if isinstance(v, dict)}
json.dump(out, open(os.path.join(os.path.dirname(src), "traffic.json"), "w"), indent=1)
print(json.dumps(out)[:300])
