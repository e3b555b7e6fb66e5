#!/bin/bash
# where the fp16 forward tile's time is: parts left out (wrong results; -DIGAN_DIAGNOSTIC build, IGAN_DIAG_MODE), whole forward / data-gradient calls of three layers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
export IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_diag.so
for m in 0 32 8 16 64 1 40; do
  for layer in "G 128 Conv1" "G 32 Conv1"; do
    IGAN_DIAG_MODE=$m timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/mode $m: /"
  done
done | tee $O/ablate.txt
