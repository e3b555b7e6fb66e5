#!/bin/bash
# A/B of one environment switch on the SAME box (boxes differ by a few percent): runs bench.py alternately with
# VAR=1 / VAR=0, twice each, and prints img/s, ms/step and the per-op device times.
# usage: tools/ab_bench.sh IGAN_STYLE_FUSION
VAR=${1:?env var}
for i in 1 2; do for f in 1 0; do
  env $VAR=$f python bench.py --no-cpu-baseline --no-roofline --op-times 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$f', d['value'], d['ms_per_step'], d['op_ms'])"
done; done
