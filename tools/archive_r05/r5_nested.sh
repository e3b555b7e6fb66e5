#!/bin/bash
# the nested form of the tap-outermost loop (libigan_hip_nested.so = -DIGAN_F16_NESTED) against the product: parity tests on the variant, then whole calls of four layers, product first and last
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5x; mkdir -p $O
L=$PWD/inclusivegan_amd/csrc
V=${1:-nested}
IGAN_LIB=$L/libigan_hip_$V.so timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 > $O/tests_$V.txt
for v in product $V product $V; do
  if [ $v = product ]; then unset IGAN_LIB; else export IGAN_LIB=$L/libigan_hip_$v.so; fi
  for layer in "G 128 Conv1" "G 32 Conv1" "G 64 Conv1" "D 64 Conv1_down" "G 64 Conv0_up" "G 16 Conv1"; do
    timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/$v: /"
  done
done | tee $O/layers_$V.txt
