#!/bin/bash
# the bench line on the final bench.py (CPU baseline at the GPU line's minibatch_gpu)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 1200 python bench.py --conv-shapes $O/conv_shapes.txt 2> $O/bench.err | tail -1 > $O/bench_n1.json
tail -c 1200 $O/bench_n1.json; tail -3 $O/bench.err
