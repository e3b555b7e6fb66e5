#!/bin/bash
# Is the replay-vs-eager mismatch of op G under the two-piece fp16 form reproducible on ONE rank, and which kernel family carries it?
mkdir -p gpurun_out; OUT=gpurun_out/f16_pairs_stress.txt; : > $OUT
run() { echo "## $*" >> $OUT; env IGAN_GRAPH_STRESS=1 "$@" timeout 900 python bench.py --data-size 1152 --no-cpu-baseline --no-variant-line --no-roofline --steps 40 --warmup 2 --revalidate-every 1 2> gpurun_out/stress.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['hip_graphs']['checks']; print(d['value'], len(c), 'checks,', sum(1 for x in c if not x['faithful']), 'unfaithful')" >> $OUT 2>&1; grep "WARNING" gpurun_out/stress.err | cut -c1-260 | sort | uniq -c | sort -rn | head -8 >> $OUT; }
run IGAN_CONV_PLANES=2
run IGAN_CONV_PLANES=2 IGAN_WGRAD_PLANES=0
run IGAN_CONV_PLANES=2 IGAN_PIECES_SHARE=0
run IGAN_CONV_PLANES=1
cat $OUT
