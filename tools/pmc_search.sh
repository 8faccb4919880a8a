#!/bin/bash
# Run ON THE GPU BOX: PMC passes of the cavity search kernel (k_inner_bnb) and the ICP loop on 64 rows of the configs[3] batch, one host thread
# (rocprofv3's launch interception crashed with several). Output: gpurun_out/pmc_search/*.  usage: tools/pmc_search.sh
set -o pipefail
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_search
mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/c4_breakdown.py 64 1 > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/mem -- python3 $R/tools/c4_breakdown.py 64 1 > $OUT/mem.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_search/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_inner_bnb" in k or "k_icp_loop" in k:
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_search/summary.txt", "w") as fo:
    for k, cs in out.items():
        line = "%s: " % k + ", ".join("%s avg %.4g (n %d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items()))
        print(line); fo.write(line + "\n")
PY
