#!/bin/bash
# usage: ab_lib.sh <variant lib path>   -- alternate product lib / variant lib, twice each
V=$1
for i in 1 2; do for f in base var; do
  if [ $f = var ]; then export IGAN_LIB=$V; else unset IGAN_LIB; fi
  python bench.py --data-size 1152 --no-cpu-baseline --no-roofline --op-times 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d.get('op_ms'))"
done; done
