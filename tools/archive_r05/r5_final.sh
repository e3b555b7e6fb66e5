#!/bin/bash
# end of round 5: the whole GPU suite on the final code + smoke, the bench line as the driver runs it (twice) with the stamped shape table, the eight-rank replay stress
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.txt 2>&1
echo "pytest rc $?" >> $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/gpu_tests.txt
timeout 900 python bench.py --conv-shapes $O/conv_shapes.txt 2> $O/bench.err | tail -1 > $O/bench_n1.json
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/driver_$i.json; done
bash tools/r5_replay_stress.sh > $O/stress.log 2>&1
tail -5 $O/gpu_tests.txt
