for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)"
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
