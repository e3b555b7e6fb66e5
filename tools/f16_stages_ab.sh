#!/bin/bash
# Round 4: three stages (product) against four (libigan_hip_st4.so, -DIGAN_F16_STAGES=4: the knob exists in commit 81fce54 only) in the fp16 form's ring: digests, per-layer times, bench A/B.
mkdir -p gpurun_out; OUT=gpurun_out/f16_stages_ab.txt; : > $OUT
V=$PWD/inclusivegan_amd/csrc/libigan_hip_st4.so
python tools/planes_digest.py > /tmp/dig_a.txt 2>/dev/null; IGAN_LIB=$V python tools/planes_digest.py > /tmp/dig_b.txt 2>/dev/null
echo "## digests product vs 4-stage build: $(diff -q /tmp/dig_a.txt /tmp/dig_b.txt > /dev/null && echo EQUAL || echo DIFFERENT)" >> $OUT
for lib in product st4 product st4; do
  if [ $lib = st4 ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  echo "# $lib: conv_layers TOTAL (fwd us TF/s | dgrad | wgrad) and the large layers" >> $OUT
  timeout 600 python tools/conv_layers.py 0.2 2>/dev/null | grep -E "TOTAL|G 32 Conv1|G 64 Conv1|G 128 Conv1|D 64 Conv1_down|G 64 Conv0_up" >> $OUT
done
B="python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --steps 48 --warmup 8"
for i in 1 2; do for lib in product st4; do
  if [ $lib = st4 ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench $lib', d['value'], d['ms_per_step'], {k:(v.get('achieved'), v.get('share_of_conv_time')) for k,v in list(r['families'].items())[:2]}, d['hip_graphs']['faithful'])" >> $OUT
done; done
cat $OUT
