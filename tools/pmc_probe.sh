#!/bin/bash
# Run ON THE GPU BOX: extra PMC passes for the cube-sweep kernel (texture-address / L1 / TLB side). usage: tools/pmc_probe.sh tag
set -o pipefail
TAG=${1:-probe}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-register > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
cd $R
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_$TAG/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "k_eval_cubes" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in acc.items():
            print(f"{k:45s} per-launch {v / max(n,1):.6g}  (n={n})")
PY
