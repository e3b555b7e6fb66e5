#!/bin/bash
# micro-variants of the tap-outermost step (tools/conv_layers.py on two layers, same box, product first and last)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5u; mkdir -p $O
L=$PWD/inclusivegan_amd/csrc
for v in product dma1 dma2 prio nolgkm product; do
  if [ $v = product ]; then unset IGAN_LIB; else export IGAN_LIB=$L/libigan_hip_$v.so; fi
  for layer in "G 128 Conv1" "G 32 Conv1" "G 64 Conv1"; do
    timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/$v: /"
  done
done | tee $O/micro.txt
