#!/bin/bash
# The whole GPU suite on the default path, smoke, one bench line.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.txt 2>&1
echo "pytest rc $?" >> $O/gpu_tests.txt
tail -30 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/gpu_tests.txt
timeout 900 python bench.py --no-variant-line > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
