#!/bin/bash
# A/B of the cavity-sized search side: tools/ab_cav.sh libgoicp_hip.so libgoicp_v1.so ...  (see tools/ab_bench.sh for building variants):
# summed Register() ms and engine-kernel ms of the 8 pairs of tools/cav_perf.py (second repeat), plus the shipped hard pair 2.
for L in "$@"; do
  GOICP_LIB=$PWD/go-icp-protein-cavities_amd/$L python tools/cav_perf.py 2>/dev/null | tail -8 | awk -v L=$L '{r+=$12; k+=$17} END {printf "%s: 8 cavity pairs: Register %.1f ms, kernels %.1f ms\n", L, r, k}'
done
