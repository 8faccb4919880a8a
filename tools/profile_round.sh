#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes of the default bench, summaries into gpurun_out/.
# usage: tools/profile_round.sh r01
set -o pipefail
TAG=${1:-r01}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/prof_$TAG/stats.log 2>&1
# counters in their own passes (no other trace domains)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-register > $R/gpurun_out/prof_$TAG/pmc_a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-register > $R/gpurun_out/prof_$TAG/pmc_b.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-register > $R/gpurun_out/prof_$TAG/pmc_c.log 2>&1
cd $R
python3 tools/summarize_profiles.py gpurun_out/prof_$TAG $TAG
