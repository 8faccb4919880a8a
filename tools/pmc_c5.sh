#!/bin/bash
# Run ON THE GPU BOX: HBM-side traffic of the streamed sweep at the configs[4] shape (50 000 points, 500^3 grid, untrimmed).
set -o pipefail
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_c5
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_c5/a -- python3 $R/bench.py --points 50000 --dt 500 --trim 0 --steps 2 --warmup 1 --no-cpu --no-register > $R/gpurun_out/pmc_c5/a.log 2>&1 || echo "pass failed"
cd $R
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_c5/a/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "k_eval_cubes" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:20s} per-launch {v / max(n,1):.6g}  (n={n})")
for f in glob.glob("gpurun_out/pmc_c5/a/**/*kernel_trace.csv", recursive=True):
    d = [ (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "k_eval_cubes" in r["Kernel_Name"]]
    if d: print("k_eval_cubes_stream launches", len(d), "avg ms", sum(d) / len(d) / 1e6)
PY
