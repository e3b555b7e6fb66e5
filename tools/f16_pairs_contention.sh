#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/f16_pairs_contention_graphs.txt; : > $OUT
CONTENTION_GRAPHS=1 IGAN_CONV_PLANES=2 timeout 1200 python tools/planes_contention.py 8 12 >> $OUT 2>&1
CONTENTION_GRAPHS=1 IGAN_CONV_PLANES=1 timeout 1200 python tools/planes_contention.py 8 12 >> $OUT 2>&1
cat $OUT
