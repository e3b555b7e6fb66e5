#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for sh in 1 0; do
IGAN_COLMAX_SHARE=$sh timeout 600 python bench.py --no-cpu-baseline --no-variant-line --data-size 1152 > $O/bench_small_$sh.json 2> $O/bench_small_$sh.err
python -c "
import json
d=json.loads(open('$O/bench_small_$sh.json').read().strip().splitlines()[-1])
print('colmax share $sh:', d['value'], d['ms_per_step'], d['roofline']['conv_family_ms_per_iteration'], d['hip_graphs']['faithful'])
for k,v in list(d['roofline']['families'].items())[:3]: print('   ', k, v['achieved'], v['share_of_conv_time'])"
done
