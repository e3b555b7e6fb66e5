#!/bin/bash
# bench A/B of the four-wave fp16 tile (IGAN_F16_W4=1, default) against the eight-wave tile (=0), alternating on one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
for i in 1 2; do for v in 0 1; do
  IGAN_F16_W4=$v timeout 600 python bench.py --data-size 1152 --no-cpu-baseline --no-roofline --no-variant-line --op-times 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w4=$v', d['value'], d['ms_per_step'], d.get('op_ms'), d['hip_graphs']['faithful'])"
done; done | tee $O/bench_w4.txt
