#!/bin/bash
# Host topology of a GPU box as the container sees it: NUMA node / local CPU list of every amdgpu device, socket layout, the affinity mask and
# (cgroup v2) the CPU quota.  Used for profiles/r04_host_threads.txt (inclusivegan_amd/hostaffinity.py has the story).
for d in /sys/class/drm/card*/device; do
  v=$(cat $d/vendor 2>/dev/null); [ "$v" = "0x1002" ] && echo "$d numa_node=$(cat $d/numa_node) local_cpulist=$(cat $d/local_cpulist)"
done
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)"
python -c "import os; print('affinity mask:', len(os.sched_getaffinity(0)), 'cpus')"
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
