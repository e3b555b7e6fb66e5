#!/bin/bash
# the four-wave tile of the fp16 form (conv_fwd_planes_w4_kernel, IGAN_F16_W4=1, the default) against the eight-wave tile (IGAN_F16_W4=0):
# parity tests on the default, then whole calls of six layers, alternating
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8 > $O/tests_w4.txt
for v in 0 1 0 1; do
  for layer in "G 128 Conv1" "G 32 Conv1" "G 64 Conv1" "D 64 Conv1_down" "G 64 Conv0_up" "G 16 Conv1" "G 8 Conv1"; do
    IGAN_F16_W4=$v timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/w4=$v: /"
  done
done | tee $O/layers_w4.txt
