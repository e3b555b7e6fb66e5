#!/bin/bash
# small layers: row thresholds of the fp16 form (forward / data gradient: IGAN_PLANES_MIN_ROWS, weight gradient: IGAN_WGRAD_PLANES_MIN_ROWS; product 2048 both)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
for t in 2048 256 2048 256; do
  for layer in "G 4x4 Conv" "G 8 Conv0_up" "G 8 Conv1" "G 16 Conv0_up" "D 16 Conv1_down" "D 8 Conv0" "D 8 Conv1_down" "VGG conv5_2"; do
    IGAN_PLANES_MIN_ROWS=$t IGAN_WGRAD_PLANES_MIN_ROWS=$t timeout 120 python tools/conv_layers.py 0.3 "$layer" 2>/dev/null | grep "^$layer" | sed "s/^/min_rows=$t: /"
  done
done | tee $O/small.txt
