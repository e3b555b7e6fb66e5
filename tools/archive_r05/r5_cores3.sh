#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
echo "#### product aggressor" | tee -a $O/cores.txt
timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
echo "#### aggressor built with -fno-slp-vectorize (no packed fp32 instructions)" | tee -a $O/cores.txt
AGGRESSOR_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_noslp.so timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
echo "#### diag aggressor, no DMA, no epilogue, no scale table (mode 112)" | tee -a $O/cores.txt
AGGRESSOR_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_diag.so AGGRESSOR_ENV_IGAN_DIAG_MODE=112 timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
