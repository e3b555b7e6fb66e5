#!/bin/bash
# usage: kprof.sh "<layer filter>"   -> per-kernel stats of tools/conv_layers.py on that layer
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
D=/tmp/kp_$$; rm -rf $D
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 $R/tools/conv_layers.py 0.1 "$1" > /dev/null 2>&1
f=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1)
echo "== $1 ($IGAN_CONV_PLANES)"
if [ -n "$f" ]; then python3 $R/tools/prof_summary.py $f ${2:-8} < /dev/null; else echo "no stats file"; fi
rm -rf $D
