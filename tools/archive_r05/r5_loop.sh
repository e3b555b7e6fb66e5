#!/bin/bash
# the main loop of conv_fwd_planes_kernel<2, true>: (1) raw rate of the fp16 / bf16 matrix instruction, (2) LDS ring of 4 stages against 3
# (libigan_hip_ns4.so = -DIGAN_F16_NSTAGE=4), product first and last, (3) parts of the loop left out in the -DIGAN_DIAGNOSTIC build
# (IGAN_DIAG_MODE bits: 32 no LDS-DMA, 128 no barrier, 256 one fold register of sixteen, 512 no matrix instructions; wrong results, timing only)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5w; mkdir -p $O
L=$PWD/inclusivegan_amd/csrc
tools/mfma_rate 2 > $O/mfma_rate.txt 2>&1
for v in product ns4 product ns4; do
  if [ $v = product ]; then unset IGAN_LIB; else export IGAN_LIB=$L/libigan_hip_$v.so; fi
  for layer in "G 128 Conv1" "G 32 Conv1" "G 64 Conv1" "D 64 Conv1_down"; do
    timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/$v: /"
  done
done | tee $O/ring.txt
export IGAN_LIB=$L/libigan_hip_diag.so
for m in 0 32 160 288 544 416 128 256 512; do
  for layer in "G 128 Conv1" "G 32 Conv1"; do
    IGAN_DIAG_MODE=$m timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/mode $m: /"
  done
done | tee $O/loop_parts.txt
