#!/bin/bash
# the fp16 form at 1024 rows in slice-outermost order on the path-length steps of iterations 9, 13, 17: does the order account for the whole effect?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
for it in 8 12 16; do
  IGAN_F16_TAP_OUTER=0 IGAN_PLANES_MIN_ROWS=1024 IGAN_WGRAD_PLANES_MIN_ROWS=1024 IGAN_TEST_TRAJECTORY_ITS=$it IGAN_TEST_GRAD_REPORT=1 timeout 200 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q 2>&1 | grep -E "GRAD-REPORT" | tail -1 | cut -c1-330 | sed "s/^/fp16_rows1024_sliceouter it $it: /"
done | tee $O/greg4e.txt
