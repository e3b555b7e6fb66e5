#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
for f in 2 1; do
  IGAN_CONV_PLANES=$f timeout 900 python tools/planes_contention.py 8 20 > $O/contention_form$f.txt 2>&1
  cat $O/contention_form$f.txt | cut -c1-400
done
IGAN_F16_TAP_OUTER=0 timeout 900 python tools/planes_contention.py 8 20 > $O/contention_form2_sliceouter.txt 2>&1
cat $O/contention_form2_sliceouter.txt | cut -c1-400
