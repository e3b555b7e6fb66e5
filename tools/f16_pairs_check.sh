#!/bin/bash
# Round 4: first runs of the two-piece fp16 variant (IGAN_CONV_PLANES=2): parity child of tests/test_gpu_planes_variant.py, rounding against fp64, per-layer times next to the bf16 form.
OUT=gpurun_out/f16_pairs_check.txt
mkdir -p gpurun_out; : > $OUT
echo "## parity (tests/test_gpu_planes_variant.py, both forms)" >> $OUT
timeout 900 python -m pytest tests/test_gpu_planes_variant.py -m gpu -x -q -k "piece_forms" 2>&1 | tail -40 >> $OUT
echo "## rounding against fp64 (tools/split_accuracy.py)" >> $OUT
for m in 0 1 2; do IGAN_CONV_PLANES=$m timeout 300 python tools/split_accuracy.py 2>&1 | sed "s/^/PLANES=$m /" | tail -6 >> $OUT; done
echo "## per layer (tools/conv_layers.py 0.2)" >> $OUT
for m in 1 2; do echo "# IGAN_CONV_PLANES=$m" >> $OUT; IGAN_CONV_PLANES=$m timeout 600 python tools/conv_layers.py 0.2 2>&1 | tail -45 >> $OUT; done
cat $OUT
