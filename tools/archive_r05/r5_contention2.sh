#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
for f in 2 1 0; do
  CONTENTION_DIFFERENT_DATA=1 IGAN_CONV_PLANES=$f timeout 900 python tools/planes_contention.py 8 20 > $O/contention_diffdata_form$f.txt 2>&1
  cat $O/contention_diffdata_form$f.txt | cut -c1-300
done
