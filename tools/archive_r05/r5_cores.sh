#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
for k in fwd2 fwd3 fwd0 wgrad2 dense; do
  timeout 200 python tools/coresidency_probe.py $k 4 4 15 2>&1 | tee -a $O/cores.txt | cut -c1-400
done
