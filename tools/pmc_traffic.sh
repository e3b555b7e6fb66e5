#!/bin/bash
# HBM-side traffic of one layer's kernels: tools/pmc_traffic.sh "<layer filter>" <kernel substring>
R=$PWD; L=$1; K=$2
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt_$c -- python3 $R/tools/conv_layers.py 0.05 "$L" > /tmp/pt_$c.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py /tmp/pt_$c $K < /dev/null
done
