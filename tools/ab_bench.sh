#!/bin/bash
# A/B of metric-kernel variants on the GPU box: tools/ab_bench.sh libgoicp_hip.so libgoicp_v1.so ...   (libraries under go-icp-protein-cavities_amd/,
# built with `make -C go-icp-protein-cavities_amd/csrc O=v1 LIB=../libgoicp_v1.so EXTRA=-D... ../libgoicp_v1.so`); prints fused / unfused ms per launch.
# The product library is not needed for this: every run names its library through GOICP_LIB.
for L in "$@"; do
  GOICP_LIB=$PWD/go-icp-protein-cavities_amd/$L python bench.py --steps 150 --no-cpu --no-register --no-stress > gpurun_out/ab_$L.log 2>&1
  echo "$L: $(grep -o 'avg_launch_ms": [0-9.]*' gpurun_out/ab_$L.log | tr '\n' ' ')"
done
