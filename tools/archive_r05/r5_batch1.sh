#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_loop_parity.py::test_config5_two_ranks_at_its_own_size_match_oracle_towers tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_ops.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt; tail -6 $O/pytest.txt
timeout 300 python tools/conv_layers.py 0.15 > $O/layers.txt 2>&1; tail -n 1 $O/layers.txt
for f in 2 0 1; do
  IGAN_CONV_PLANES=$f timeout 900 python tools/long_ab.py --iters 3000 --out $O/long_form$f.json > $O/long_form$f.log 2>&1
  tail -2 $O/long_form$f.log
done
python tools/long_ab.py --compare $O/long_form2.json $O/long_form0.json $O/long_form1.json > $O/long_ab.txt 2>&1
head -40 $O/long_ab.txt
