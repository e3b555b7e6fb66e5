#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/f16_pairs_retest.txt; : > $OUT
for m in 2 1 2; do
  echo "## IGAN_CONV_PLANES=$m: 8-rank bench" >> $OUT
  IGAN_CONV_PLANES=$m timeout 900 python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 2> gpurun_out/retest_$m.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['hip_graphs'])" >> $OUT 2>&1
  grep -i "WARNING\|does not reproduce" gpurun_out/retest_$m.err | head -5 | cut -c1-600 >> $OUT
done
for m in 1 2; do
  echo "## IGAN_CONV_PLANES=$m: config 5 own size" >> $OUT
  IGAN_CONV_PLANES=$m timeout 900 python -m pytest tests/test_gpu_loop_parity.py -m gpu -q -s -k "config5_at_its_own_size" 2>&1 | grep -i "pl_mean\|passed\|failed\|worst\|AssertionError" | head -10 | cut -c1-400 >> $OUT
done
cat $OUT
