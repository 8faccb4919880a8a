echo "affinity: $(python3 -c 'import os; a=sorted(os.sched_getaffinity(0)); print(len(a), a[:4], a[-4:])')"
for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d numa $(cat $d/numa_node) vendor $(cat $d/vendor) local_cpus $(cat $d/local_cpulist)"; done 2>/dev/null | head -12
for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
