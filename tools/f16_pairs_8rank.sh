#!/bin/bash
# Eight ranks on ONE GPU over gloo at the bench size (the replay-vs-eager check of every captured op under multi-process load), once per argument;
# an argument is a space-separated list of environment assignments for that run ("" = the default path).  This is the harness behind sections 5, 8 and 11 of
# profiles/r04_f16_pairs_variant.txt (builds that no longer exist were selected with IGAN_LIB=<variant .so>).
# usage: tools/f16_pairs_8rank.sh "IGAN_CONV_PLANES=2" "IGAN_CONV_PLANES=2 IGAN_WGRAD_PLANES=0" "IGAN_CONV_PLANES=1"
mkdir -p gpurun_out; OUT=gpurun_out/f16_pairs_8rank.txt; : > $OUT
for cfg in "$@"; do
  echo "## ${cfg:-default}" >> $OUT
  env $cfg timeout 900 python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 2> gpurun_out/8rank.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['hip_graphs']['checks'])" >> $OUT 2>&1
  grep -i "does not reproduce" gpurun_out/8rank.err | cut -c1-200 | sort | uniq -c | sort -rn | head -6 >> $OUT
done
cat $OUT
