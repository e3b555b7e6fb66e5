#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/f16_pairs_8rank6.txt; : > $OUT
run() { echo "## $*" >> $OUT; env "$@" timeout 900 python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 2> gpurun_out/8rank.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['hip_graphs']['checks'])" >> $OUT 2>&1; grep -i "does not reproduce" gpurun_out/8rank.err | cut -c1-60 | sort | uniq -c | sort -rn | head -4 >> $OUT; }
V=$PWD/inclusivegan_amd/csrc/libigan_hip_builtindma.so
run IGAN_CONV_PLANES=2 IGAN_LIB=$V
run IGAN_CONV_PLANES=2 IGAN_LIB=$V
run IGAN_CONV_PLANES=2 IGAN_LIB=$V
run IGAN_CONV_PLANES=2
run IGAN_CONV_PLANES=2
for m in 1 2; do
  echo "## IGAN_CONV_PLANES=$m: config 5 own size" >> $OUT
  IGAN_CONV_PLANES=$m timeout 900 python -m pytest tests/test_gpu_loop_parity.py -m gpu -q -s -k "config5_at_its_own_size" 2>&1 | grep -i "worst\|passed\|failed\|AssertionError" | head -5 | cut -c1-400 >> $OUT
done
cat $OUT
