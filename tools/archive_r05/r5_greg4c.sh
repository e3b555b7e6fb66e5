#!/bin/bash
# path-length steps of iterations 1, 9, 13, 17 (0-based 0, 8, 12, 16) against the fp64 oracle under the two row thresholds: is the larger deviation at iteration 5 systematic?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
run() {  # label, its, env...
  label=$1; its=$2; shift; shift
  env "$@" IGAN_TEST_TRAJECTORY_ITS=$its IGAN_TEST_GRAD_REPORT=1 timeout 900 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q 2>&1 | grep -E "GRAD-REPORT" | tail -1 | cut -c1-330 | sed "s/^/$label it $its: /"
}
{
for it in 0 8 12 16; do
  run rows1024 $it IGAN_NOOP=1
  run rows2048 $it IGAN_PLANES_MIN_ROWS=2048 IGAN_WGRAD_PLANES_MIN_ROWS=2048
done
} | tee $O/greg4c.txt
