#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
L=$PWD/inclusivegan_amd/csrc
echo "#### aggressor with its LDS padded to 75264 B (the bf16 tile's footprint)" | tee -a $O/cores.txt
AGGRESSOR_LIB=$L/libigan_hip_pad.so timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
echo "#### victim: transposed dense form loads its weights 4 bytes at a time" | tee -a $O/cores.txt
VICTIM_LIB=$L/libigan_hip_wtb32.so timeout 120 python tools/coresidency_probe.py fwd2 2 4 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
echo "#### one aggressor only" | tee -a $O/cores.txt
timeout 120 python tools/coresidency_probe.py fwd2 2 1 8 2>&1 | tee -a $O/cores.txt | grep -v AGGRESSOR | cut -c1-330
