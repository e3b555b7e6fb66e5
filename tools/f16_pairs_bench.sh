#!/bin/bash
# Same-box A/B of the training step: bf16 x3 (default) vs the two-piece fp16 variant (IGAN_CONV_PLANES=2), alternating.
OUT=gpurun_out/f16_pairs_bench.txt
mkdir -p gpurun_out; : > $OUT
B="python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --steps 48 --warmup 8 --op-times"
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1', d['value'], d['ms_per_step'], 'dev', d.get('op_ms'), 'faithful', d['hip_graphs'].get('faithful'), 'families', {k:(v.get('achieved'), v.get('share_of_conv_time')) for k,v in list(r.get('families',{}).items())[:5]}, 'conv ms/it', r.get('conv_family_ms_per_iteration'))
except Exception as e: print('$1', 'FAILED', repr(e))"; }
for i in 1 2; do for m in 1 2; do
  IGAN_CONV_PLANES=$m timeout 900 $B --conv-shapes gpurun_out/f16_pairs_shapes_$m.txt 2>gpurun_out/f16_pairs_bench_$m.err | line "PLANES=$m" >> $OUT
done; done
cat $OUT
