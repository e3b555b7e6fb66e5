#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_planes_variant.py -m gpu -x -q 2>&1 | tail -15 > $O/variant_tests.txt
run() {  # label, env...
  label=$1; shift
  env "$@" IGAN_TEST_TRAJECTORY_ITS=4 IGAN_TEST_GRAD_REPORT=1 timeout 900 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q 2>&1 | grep -E "GRAD-REPORT" | tail -1 | sed "s/^/$label: /"
}
{
run fwd1024_wgrad2048 IGAN_PLANES_MIN_ROWS=1024 IGAN_WGRAD_PLANES_MIN_ROWS=2048
run fwd2048_wgrad1024 IGAN_PLANES_MIN_ROWS=2048 IGAN_WGRAD_PLANES_MIN_ROWS=1024
run both1024_nocolmaxshare IGAN_COLMAX_SHARE=0
} | tee $O/greg4b.txt
