#!/bin/bash
# end of round 5, after the row thresholds went back to 2048: the whole GPU suite + smoke, then the counter passes and the evidence set on the final sources
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5f2; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.txt 2>&1
echo "pytest rc $?" >> $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/gpu_tests.txt
tail -4 $O/gpu_tests.txt
bash tools/r5_evidence.sh > $O/evidence.log 2>&1
tail -c 600 gpurun_out/prof_r05/bench_n1.json
