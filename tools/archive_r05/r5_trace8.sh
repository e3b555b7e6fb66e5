#!/bin/bash
# localising the eight-process nondeterminism of the fp16 form: per-Function checksums of two eager executions; a build without the window counter's atomics; the kernels alone with DIFFERENT data per process
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
make -C inclusivegan_amd/csrc variant VARIANT=nowin DEFS=-DIGAN_NO_WINDOW_COUNT > $O/build.txt 2>&1
B="python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1"
verdict() { python -c "
import sys, json
lines = [l for l in open('$1').read().splitlines() if l.startswith('{')]
print('$2', 'faithful', json.loads(lines[-1])['hip_graphs']['faithful'] if lines else 'NO RESULT')"; }
for i in 1 2; do
  IGAN_GRAPH_CHECK_EAGER_TWICE=1 IGAN_GRAPH_CHECK_TRACE=1 IGAN_GRAPH_CHECK_VERBOSE=1 timeout 200 $B > $O/out_trace_$i.txt 2> $O/err_trace_$i.txt
  verdict $O/out_trace_$i.txt "trace run $i"
  grep -h "TRACE-DIFF" $O/out_trace_$i.txt $O/err_trace_$i.txt | cut -c1-330 | head -40
done
for i in 1 2 3; do
  IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_nowin.so IGAN_GRAPH_CHECK_EAGER_TWICE=1 timeout 200 $B > $O/out_nowin_$i.txt 2> $O/err_nowin_$i.txt
  verdict $O/out_nowin_$i.txt "no window-counter atomics run $i"
done
CONTENTION_DIFFERENT_DATA=1 IGAN_CONV_PLANES=2 timeout 400 python tools/planes_contention.py 8 12 > $O/contention_diffdata_form2.txt 2>&1
cat $O/contention_diffdata_form2.txt | cut -c1-300
