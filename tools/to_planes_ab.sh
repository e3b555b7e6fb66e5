#!/bin/bash
# Round 4: to_planes_f16_kernel rewritten (lane-contiguous float4 loads, quad shuffles, 16 B stores, fixed grid: block maxima read once per block) against the first version
# (libigan_hip_oldimg.so = the evidence build; the rewritten kernel itself is in commit ed78f43, not in the tree): digests, piece / op tests, image times, bench A/B.
mkdir -p gpurun_out; OUT=gpurun_out/to_planes_ab.txt; : > $OUT
V=$PWD/inclusivegan_amd/csrc/libigan_hip_oldimg.so
python tools/planes_digest.py > /tmp/dig_a.txt 2>/dev/null; IGAN_LIB=$V python tools/planes_digest.py > /tmp/dig_b.txt 2>/dev/null
echo "## digests new vs first image kernel: $(diff -q /tmp/dig_a.txt /tmp/dig_b.txt > /dev/null && echo EQUAL || echo DIFFERENT)" >> $OUT
timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -2 >> $OUT
B="python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --steps 48 --warmup 8"
for i in 1 2; do for lib in new old; do
  if [ $lib = old ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; f=r['families']; print('bench $lib', d['value'], d['ms_per_step'], 'images:', f['to_planes_kernel'], 'conv ms/it', r['conv_family_ms_per_iteration'], d['hip_graphs']['faithful'], d['fp16_pairs_window']['fraction'])" >> $OUT
done; done
cat $OUT
