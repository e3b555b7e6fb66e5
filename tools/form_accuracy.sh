#!/bin/bash
# Loss values and gradients of the four training phases at 128x128 config-e against the fp64 oracle, per convolution form (0 fp32 instruction, 1 bf16 x3, 2 fp16 x2)
mkdir -p gpurun_out; OUT=gpurun_out/form_accuracy.txt; : > $OUT
for m in 0 1 2; do
  echo "## IGAN_CONV_PLANES=$m" >> $OUT
  IGAN_CONV_PLANES=$m timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q -s -k "at_128_config_e and config4" 2>&1 | grep -i "worst per-variable\|passed\|failed\|Error" | head -5 | cut -c1-600 >> $OUT
done
cat $OUT
