#!/bin/bash
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_v
IGAN_CONV_PLANES=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_v -o kt -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-variant-line --steps 24 --data-size 1152 > /dev/null 2>&1 < /dev/null
f=$(find /tmp/kt_v -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python3 $R/tools/gpu_idle.py $f 0.3 12 < /dev/null; else echo "no trace"; fi
