#!/bin/bash
# Timing-only ablations of conv_fwd_planes_kernel (variant libraries built with -DIGAN_PLANES_NO_*: WRONG results, never the product):
# which part of the step does the time go to?  tools/planes_ablate.sh [seconds]
# Build first:  for v in nofold:NO_FOLD noprep:NO_PREP noldsread:NO_LDSREAD nodmaa:NO_DMA_A; do make -C inclusivegan_amd/csrc variant VARIANT=${v%%:*} DEFS=-DIGAN_PLANES_${v##*:}; done
S=${1:-0.3}
for L in "G 32 Conv1" "G 128 Conv1"; do
  for v in product nofold noprep noldsread nodmaa; do
    if [ $v = product ]; then unset IGAN_LIB; else export IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_$v.so; fi
    echo "$v: $(python tools/conv_layers.py $S "$L" 2>/dev/null | grep "$L")"
  done
done
