#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + PMC passes of bench.py, summaries into gpurun_out/prof_<tag>[_c5].
# usage: tools/profile_round.sh r02            (default shape = BASELINE configs[2])
#        tools/profile_round.sh r02 c5         (configs[4] shape: 50 000 points, 500^3 grid, untrimmed)
# Counters are collected in passes of their own with --kernel-trace only (no other trace domains), each within the per-block slot
# limits of /opt/skills/guides/MI355X_MICROARCH.md (TCC: 4 slots, FETCH_SIZE = 3, WRITE_SIZE = 2; SQ: 8).
set -o pipefail
TAG=${1:-r03}
SHAPE=${2:-c3}
R=$PWD
export TMPDIR=/tmp
ARGS=""
OUT=$R/gpurun_out/prof_$TAG
if [ "$SHAPE" = "c5" ]; then ARGS="--points 50000 --dt 500 --trim 0"; OUT=$R/gpurun_out/prof_${TAG}_c5; fi
mkdir -p $OUT
cd /tmp
if [ "$SHAPE" = "c5" ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS --steps 5 --warmup 1 --no-cpu --no-register > $OUT/stats.log 2>&1
else
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu --cavity-threads 1 --cavity-rows 64 --overlap-pairs 1 > $OUT/stats.log 2>&1     # (one host thread: rocprofv3 crashed in its launch interception with eight)
fi
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pmc_a -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-register > $OUT/pmc_a.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_b -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-register > $OUT/pmc_b.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_c -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-register > $OUT/pmc_c.log 2>&1
cd $R
python3 tools/summarize_profiles.py $OUT ${TAG}$( [ "$SHAPE" = "c5" ] && echo _c5 )
