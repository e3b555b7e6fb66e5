#!/bin/bash
# the path-length step of iteration 5 (0-based 4) of the config-2 run against the fp64 oracle under each convolution form / tile / threshold: largest per-variable gradient deviations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
run() {  # label, env...
  label=$1; shift
  env "$@" IGAN_TEST_TRAJECTORY_ITS=4,8 IGAN_TEST_GRAD_REPORT=1 timeout 900 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q 2>&1 | grep -E "GRAD-REPORT|passed|failed|worst" | tail -3 | sed "s/^/$label: /"
}
{
run default IGAN_NOOP=1
run w8 IGAN_F16_W4=0
run rows2048 IGAN_PLANES_MIN_ROWS=2048 IGAN_WGRAD_PLANES_MIN_ROWS=2048
run bf16x3 IGAN_CONV_PLANES=1
run fp32 IGAN_CONV_PLANES=0
} | tee $O/greg4.txt
