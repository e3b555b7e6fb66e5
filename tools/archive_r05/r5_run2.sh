#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py -x -q -s > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for m in 0 1; do IGAN_F16_TAP_OUTER=$m timeout 600 python tools/conv_layers.py 0.15 > $O/layers_tapo$m.txt 2>&1; tail -n 1 $O/layers_tapo$m.txt; done
for l in "G 128 Conv1" "G 32 Conv1"; do
  for m in 0 1; do echo "--- tap outer $m"; IGAN_F16_TAP_OUTER=$m bash tools/kprof.sh "$l" 12; done
done > $O/kprof.txt 2>&1
cat $O/kprof.txt | cut -c1-200 | head -120
