#!/bin/bash
# Round 4: the weight-gradient kernel's LDS-DMA issued from inline assembly (product) against the builtin (libigan_hip_builtindma.so, -DIGAN_WGRAD_ASM_DMA=0):
# digests (must be equal), parity, per-layer times, both piece forms.
mkdir -p gpurun_out; OUT=gpurun_out/wgrad_asm_dma_ab.txt; : > $OUT
V=$PWD/inclusivegan_amd/csrc/libigan_hip_builtindma.so
for m in 1 2; do
  IGAN_CONV_PLANES=$m python tools/planes_digest.py > /tmp/dig_a_$m.txt 2>/dev/null
  IGAN_CONV_PLANES=$m IGAN_LIB=$V python tools/planes_digest.py > /tmp/dig_b_$m.txt 2>/dev/null
  echo "## form $m: digests product vs builtin-DMA build: $(diff -q /tmp/dig_a_$m.txt /tmp/dig_b_$m.txt > /dev/null && echo EQUAL || echo DIFFERENT)" >> $OUT
  cat /tmp/dig_a_$m.txt >> $OUT
done
echo "## parity" >> $OUT
timeout 900 python -m pytest tests/test_gpu_planes_variant.py -m gpu -q 2>&1 | tail -3 >> $OUT
for m in 1 2; do for lib in product builtin; do
  echo "# form $m, $lib DMA: tools/conv_layers.py 0.2 (wgrad column = whole call incl. both piece images)" >> $OUT
  if [ $lib = builtin ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  IGAN_CONV_PLANES=$m timeout 600 python tools/conv_layers.py 0.2 2>/dev/null | awk '{print $1, $2, $3, $(NF-1), $NF}' | tail -34 | tr '\n' ';' | fold -w 2000 >> $OUT; echo >> $OUT
done; done
unset IGAN_LIB
B="python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --steps 48 --warmup 8 --op-times"
for i in 1 2; do for lib in product builtin; do
  if [ $lib = builtin ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench form 1 $lib DMA', d['value'], d['ms_per_step'], d['op_ms'], {k:(v.get('achieved'), v.get('share_of_conv_time')) for k,v in list(r['families'].items())[:3]})" >> $OUT
done; done
cat $OUT
