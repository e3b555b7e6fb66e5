#!/bin/bash
# bench with the fp16 form's row thresholds lowered (IGAN_PLANES_MIN_ROWS / IGAN_WGRAD_PLANES_MIN_ROWS), alternating on one box; first the parity tests at the lowest setting
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
IGAN_PLANES_MIN_ROWS=512 IGAN_WGRAD_PLANES_MIN_ROWS=512 timeout 900 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py tests/test_gpu_ops.py tests/test_gpu_networks.py -m gpu -x -q 2>&1 | tail -4 > $O/tests_small.txt
for i in 1 2; do for t in "2048 2048" "1024 1024" "512 512" "1024 512" "2048 512"; do
  set -- $t
  IGAN_PLANES_MIN_ROWS=$1 IGAN_WGRAD_PLANES_MIN_ROWS=$2 timeout 600 python bench.py --data-size 1152 --no-cpu-baseline --no-roofline --no-variant-line --op-times 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd_min=$1 wgrad_min=$2', d['value'], d['ms_per_step'], d.get('op_ms'), d['hip_graphs']['faithful'])"
done; done | tee $O/bench_small.txt
