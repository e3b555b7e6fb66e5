#!/bin/bash
# the ops of iterations 1-9 of the config-2 run (22 ops incl. three path-length steps and one R1 step) against the fp64 oracle on the final default path
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
IGAN_TEST_TRAJECTORY_ITS=0,1,2,3,4,5,6,7,8 timeout 780 python -m pytest tests/test_gpu_loop_parity.py -m gpu -k config2 -s -q > $O/trajectory_part.txt 2>&1
tail -4 $O/trajectory_part.txt
