#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
B="python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1"
for i in 1 2; do
  IGAN_GRAPH_CHECK_EAGER_TWICE=1 IGAN_GRAPH_CHECK_TRACE=1 timeout 200 $B > $O/out_trace_$i.txt 2> $O/err_trace_$i.txt
  grep -h "TRACE-DIFF" $O/out_trace_$i.txt $O/err_trace_$i.txt | cut -c1-300 | sort -t' ' -k5,5n | head -30
  echo ----
done
