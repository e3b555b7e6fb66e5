#!/bin/bash
# all 47 ops of the config-2 run against the fp64 oracle on the final default path (per-pixel / per-channel scales, four-wave tile)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
IGAN_TEST_TRAJECTORY_ALL=1 timeout 2400 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q > $O/trajectory_all.txt 2>&1
tail -5 $O/trajectory_all.txt
