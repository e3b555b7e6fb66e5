#!/bin/bash
# The driver's command twice in a row, then a sustained run (200 iterations) of the default path.
mkdir -p gpurun_out/r04
for i in 1 2; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r04/driver_cmd_$i.err | tail -1 > gpurun_out/r04/driver_cmd_$i.json
  python -c "import json; d=json.loads(open('gpurun_out/r04/driver_cmd_$i.json').read()); print('driver command run $i:', d['value'], d['ms_per_step'], d['imle_refresh_s'], d['roofline']['frac'], d['second_line_exact_fp32']['value'], d['line_bf16_pieces']['value'], d['hip_graphs']['faithful'], d['roofline'].get('traffic'))"
done
python bench.py --steps 200 --warmup 40 --data-size 1152 --no-cpu-baseline --no-variant-line 2>/dev/null | tail -1 > gpurun_out/r04/bench_sustained_200_fp16.json
python -c "import json; d=json.loads(open('gpurun_out/r04/bench_sustained_200_fp16.json').read()); print('sustained 200:', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['fp16_pairs_window'])"
