#!/bin/bash
# is the path-length step's sensitivity to the 1024-row threshold a property of the fp16 arithmetic, or of routing those calls through the piece path at all?  the exact bf16-piece form at 1024 rows
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
run() {  # label, env...
  label=$1; shift
  env "$@" IGAN_TEST_TRAJECTORY_ITS=4 IGAN_TEST_GRAD_REPORT=1 timeout 400 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q 2>&1 | grep -E "GRAD-REPORT" | tail -1 | cut -c1-420 | sed "s/^/$label: /"
}
{
run bf16x3_rows1024 IGAN_CONV_PLANES=1 IGAN_PLANES_MIN_ROWS=1024 IGAN_WGRAD_PLANES_MIN_ROWS=1024
run fp16_rows1024_sliceouter IGAN_F16_TAP_OUTER=0 IGAN_PLANES_MIN_ROWS=1024 IGAN_WGRAD_PLANES_MIN_ROWS=1024
} | tee $O/greg4d.txt
