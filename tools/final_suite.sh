#!/bin/bash
# The whole GPU suite on the default path, smoke, and three extra 8-ranks-on-one-GPU bench runs (replay checks).
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r04/gpu_tests_s2.txt 2>&1
tail -25 gpurun_out/r04/gpu_tests_s2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r04/gpu_tests_s2.txt
for i in 1 2 3; do
  timeout 900 python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 ranks on one GPU:', d['value'], d['dtype'][:40], d['hip_graphs']['checks'])" | tee -a gpurun_out/r04/gpu_tests_s2.txt
done
