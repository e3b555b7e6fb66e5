#!/bin/bash
# per-kernel stats of a short bench run: tools/kprof_bench.sh [lines]    (environment switches are inherited)
R=$PWD; cd /tmp; export TMPDIR=/tmp
D=/tmp/kb_$$; rm -rf $D
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 $R/bench.py --data-size 1152 --no-cpu-baseline --no-roofline --no-variant-line --steps 48 > /dev/null 2>&1 < /dev/null
f=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then python3 $R/tools/prof_summary.py $f ${1:-30} < /dev/null; else echo "no stats file"; fi
rm -rf $D
