#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
export IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_diag.so
for m in 0 1 2 4 8 16 32 64 40; do
  echo "#### IGAN_DIAG_MODE=$m" | tee -a $O/cores.txt
  IGAN_DIAG_MODE=$m timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
done
echo "#### slice outer" | tee -a $O/cores.txt
IGAN_F16_TAP_OUTER=0 timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
