#!/bin/bash
# The whole GPU suite under the two-piece fp16 form, then a bench run of it with its window counters.
mkdir -p gpurun_out
IGAN_CONV_PLANES=2 timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/f16_pairs_suite.txt 2>&1
tail -30 gpurun_out/f16_pairs_suite.txt
IGAN_CONV_PLANES=2 timeout 600 python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --steps 100 --warmup 20 > gpurun_out/f16_pairs_bench100.json 2> gpurun_out/f16_pairs_bench100.err
tail -c 3000 gpurun_out/f16_pairs_bench100.json
