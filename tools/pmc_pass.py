"""Run rocprofv3 --pmc passes (one per counter set; SQ / TCC / TCP slots are limited, MI355X_MICROARCH.md "rocprofv3 PMC
slots") over a command and print per-kernel averages per dispatch.

usage: python tools/pmc_pass.py OUT.json [--group NAME=REGEX ...] -- <command ...>
Counter sets are fixed below (L2 hit/miss, fabric requests, L1->L2 traffic by kind, SQ issue / wait)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

SETS = [
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    ["TCC_ATOMIC_sum", "TCC_WRITE_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES",
     "SQ_BUSY_CYCLES"],
    ["SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM",
     "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    ["FETCH_SIZE"], ["WRITE_SIZE"],
]


def main():
    argv = sys.argv[1:]
    out_path = argv[0]
    sep = argv.index("--")
    groups = [a.split("=", 1)[1].split("=", 1) if False else a[len("--group "):] for a in []]
    groups = []
    i = 1
    while i < sep:
        if argv[i] == "--group":
            name, rx = argv[i + 1].split("=", 1)
            groups.append((name, rx))
            i += 2
        else:
            i += 1
    cmd = [os.path.abspath(a) if os.path.exists(a) else a for a in argv[sep + 1:]]   # the passes run from /tmp
    res = collections.defaultdict(dict)
    tmp = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k, ctrs in enumerate(SETS):
        d = os.path.join(tmp, f"p{k}")
        r = subprocess.run(["rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd],
                           cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {k} {ctrs} failed: {r.stderr[-400:]}", file=sys.stderr)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                kn = row["Kernel_Name"]
                name = None
                for gname, rx in groups:
                    if re.search(rx, kn):
                        name = gname
                        break
                if name is None:
                    if groups:
                        continue
                    name = kn[:80]
                a = acc[name][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
        for name, cs in acc.items():
            for c, (tot, n) in cs.items():
                res[name][c] = tot / n
                res[name]["dispatches"] = n
    shutil.rmtree(tmp, ignore_errors=True)
    for name, d in res.items():
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            d["L2_hit_rate"] = round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1), 4)
        if "FETCH_SIZE" in d:
            d["fetch_bytes_x2"] = 2 * d["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in d:
            d["write_bytes"] = d["WRITE_SIZE"] * 1024
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for name, d in sorted(res.items()):
        print(name, json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in sorted(d.items())}))


if __name__ == "__main__":
    main()
