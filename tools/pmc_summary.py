"""Per-kernel averages of the rocprofv3 --pmc passes written by tools/prof_*.sh.

HBM bytes per dispatch, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request of a wide
coalesced stream, so the read side is doubled (`fetch_x2`)."""
import csv, glob, json, sys, collections

out = sys.argv[1]
res = collections.defaultdict(dict)
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    for k, (s, n) in acc.items():
        res[k][name] = s / n
        res[k]["dispatches_" + name] = n
rows = []
for k, d in res.items():
    f = d.get("FETCH_SIZE", 0.0) * 1024
    w = d.get("WRITE_SIZE", 0.0) * 1024
    rows.append({"kernel": k[:160], "fetch_bytes_raw": round(f), "fetch_bytes_x2": round(2 * f), "write_bytes": round(w),
                 "hbm_bytes_per_dispatch": round(2 * f + w), "dispatches": d.get("dispatches_FETCH_SIZE", 0)})
rows.sort(key=lambda r: -r["hbm_bytes_per_dispatch"])
for r in rows[:12]:
    print(json.dumps(r))
json.dump(rows, open(f"{out}/pmc_summary.json", "w"), indent=1)
