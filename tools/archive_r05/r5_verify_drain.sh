#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
B="python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1"
for i in 1 2 3 4 5; do
  timeout 200 $B > $O/out_$i.txt 2> $O/err_$i.txt
  python -c "
import json
lines=[l for l in open('$O/out_$i.txt').read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1]) if lines else None
print('8 ranks run $i:', (d['value'], d['hip_graphs']['faithful'], d['hip_graphs']['checks']) if d else 'NO RESULT')"
done
timeout 1500 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py "tests/test_gpu_dist.py::test_eight_rank_replay_stress" "tests/test_gpu_dist.py::test_non_finite_gradient_on_one_rank_stops_every_rank" "tests/test_gpu_dist.py::test_bench_eight_ranks_at_the_bench_size" tests/test_gpu_loop_parity.py::test_config5_two_ranks_at_its_own_size_match_oracle_towers -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
