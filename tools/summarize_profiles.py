#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into <dir>/<tag>_summary.json + <tag>_kernel_stats.csv
(the files that get copied into profiles/)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

d, tag = sys.argv[1], sys.argv[2]
out = {"tag": tag, "kernels": {}, "pmc": {}}
st = glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv"))
if st:
    shutil.copyfile(st[0], os.path.join(d, "%s_kernel_stats.csv" % tag))
    for r in csv.DictReader(open(st[0])):
        name = r["Name"].split("(")[0]
        out["kernels"][name] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), total_ns=float(r["TotalDurationNs"]), pct=float(r["Percentage"]))
for sub in ("pmc_a", "pmc_b", "pmc_c"):
    for f in glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "k_eval_cubes" in k or "k_inner_bnb" in k:
                out["pmc"].setdefault(k, {})
                for c, v in cs.items():
                    out["pmc"][k][c] = dict(n=len(v), avg=sum(v) / len(v))
# HBM/fabric bytes per launch of the metric kernel, per the guide: FETCH_SIZE is in KB and counts 64-B requests
out["hbm_bytes_per_launch_by_kernel"] = {}
for k, cs in out["pmc"].items():
    if "k_eval_cubes" in k and "FETCH_SIZE" in cs:
        out["hbm_bytes_per_launch_by_kernel"][k.replace("void ", "").replace(", ", ",")] = (cs["FETCH_SIZE"]["avg"] + cs.get("WRITE_SIZE", {"avg": 0})["avg"]) * 1024.0
for k, cs in out["pmc"].items():
    if "k_eval_cubes" in k and "FETCH_SIZE" in cs:
        fetch = cs["FETCH_SIZE"]["avg"] * 1024.0
        write = cs.get("WRITE_SIZE", {"avg": 0})["avg"] * 1024.0
        out["hbm_bytes_per_launch"] = fetch + write
        out["kernel"] = k.replace("void ", "").replace(", ", ",")
        hit = cs.get("TCC_HIT_sum", {}).get("avg"); miss = cs.get("TCC_MISS_sum", {}).get("avg")
        if hit is not None and miss is not None and hit + miss > 0:
            out["l2_hit_rate"] = hit / (hit + miss)
        out["fetch_note"] = ("FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch, from separate --pmc passes. The reads are 2- and 4-byte gathers, i.e. 64-B "
                             "requests, so the guide's x2 correction for wide 128-B streaming reads does not apply. These are fabric-side bytes (L2 misses): "
                             "a 300^3 grid (54 MB coded + 108 MB float) is Infinity-Cache resident, so they bound HBM-device traffic from above; "
                             "a 500^3 grid (250 + 500 MB) is not.")
json.dump(out, open(os.path.join(d, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(out)[:1500])
