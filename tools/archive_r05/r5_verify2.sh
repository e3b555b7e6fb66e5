#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
B="python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1"
for i in 1 2 3 4; do
  IGAN_GRAPH_CHECK_EAGER_TWICE=1 IGAN_GRAPH_CHECK_TRACE=1 timeout 200 $B > $O/out_$i.txt 2> $O/err_$i.txt
  python -c "
import json
lines=[l for l in open('$O/out_$i.txt').read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1]) if lines else None
print('8 ranks (eager twice) run $i:', (d['value'], d['hip_graphs']['faithful']) if d else 'NO RESULT')"
  grep -h "TRACE-DIFF" $O/out_$i.txt $O/err_$i.txt | cut -c1-260 | sort -t' ' -k5,5n | head -6
done
