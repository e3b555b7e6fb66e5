#!/bin/bash
# Same-box A/B of two library builds on the piece-form layers: digests must be equal, then per-layer times alternate (tools/conv_layers.py).
#   tools/planes_sched_ab.sh <variant .so> [seconds per measurement]
V=$(realpath $1); S=${2:-0.3}
python tools/planes_digest.py > /tmp/dig_a.txt; IGAN_LIB=$V python tools/planes_digest.py > /tmp/dig_b.txt
if diff /tmp/dig_a.txt /tmp/dig_b.txt > /dev/null; then echo "DIGESTS EQUAL"; else echo "DIGESTS DIFFER"; diff /tmp/dig_a.txt /tmp/dig_b.txt; fi
cat /tmp/dig_a.txt
for i in 1 2; do
  for L in "G 32 Conv1" "G 64 Conv1" "G 128 Conv1" "G 64 Conv0_up" "D 64 Conv1_down" "G 16 Conv1"; do
    echo "product: $(python tools/conv_layers.py $S "$L" 2>/dev/null | tail -1)"
    echo "variant: $(IGAN_LIB=$V python tools/conv_layers.py $S "$L" 2>/dev/null | tail -1)"
  done
done
