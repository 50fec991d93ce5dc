#!/usr/bin/env python3
"""Summarise FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc` into per-kernel HBM traffic per launch.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads -- MI355X_MICROARCH.md,
"HBM"); both counters are in KB."""
import collections, csv, json, re, sys


def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        a = agg[k]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in f:
    if k in w and f[k][1] == w[k][1]:
        fetch = 2.0 * f[k][0] / f[k][1] * 1024
        write = w[k][0] / w[k][1] * 1024
        out[k] = {"launches": f[k][1], "fetch_bytes_x2": round(fetch), "write_bytes": round(write),
                  "hbm_bytes_per_launch": round(fetch + write)}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])), open(sys.argv[3], "w"), indent=1)
