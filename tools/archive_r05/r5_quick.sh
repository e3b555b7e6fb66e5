#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_fullsize.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/conv_layers.py 0.15 > $O/layers.txt 2>&1; tail -n 1 $O/layers.txt
timeout 600 python bench.py --no-cpu-baseline --no-variant-line --data-size 1152 --conv-shapes $O/conv_shapes.txt > $O/bench_small.json 2> $O/bench_small.err
python -c "
import json
d=json.loads(open('$O/bench_small.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['conv_family_ms_per_iteration'], d['fp16_pairs_window'])
for k,v in d['roofline']['families'].items(): print(k, v['achieved'], v['share_of_conv_time'])"
