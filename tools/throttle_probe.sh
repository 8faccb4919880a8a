#!/bin/bash
# GPU box: configs[3] batch rate against the number of host threads (= registrations in flight), with the cgroup's CPU-throttling counters
# around each run (the box gives a GPU 16 CPUs' worth of CFS quota, not a CPU set: 16 spinning threads plus the runtime's helpers run into it).
for T in "$@"; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  python bench.py --no-cpu --no-stress --overlap-pairs 1 --steps 20 --cavity-threads $T > gpurun_out/throttle_$T.log 2>&1
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
  echo "threads $T: $(grep -o '"cavity_pair_registrations_per_sec": [0-9.]*' gpurun_out/throttle_$T.log) | throttled before [$a] after [$b]"
done
