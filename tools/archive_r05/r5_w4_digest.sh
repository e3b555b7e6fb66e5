#!/bin/bash
# bit-equality of the four-wave tile (IGAN_F16_W4=1) and the eight-wave tile (=0): SHA-1 digests of forward / data gradient / weight gradient on five shapes (tools/planes_digest.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
IGAN_F16_W4=0 python tools/planes_digest.py > $O/digest_w8.txt 2>/dev/null
IGAN_F16_W4=1 python tools/planes_digest.py > $O/digest_w4.txt 2>/dev/null
if diff $O/digest_w8.txt $O/digest_w4.txt > $O/digest_diff.txt; then echo "IDENTICAL"; else echo "DIFFERENT"; fi | tee -a $O/digest_diff.txt
cat $O/digest_w4.txt
