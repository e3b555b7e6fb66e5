#!/bin/bash
# Bisecting the eight-ranks-on-one-GPU replay mismatch: bench.py --gpus 8 --one-gpu under several switches, a few runs each; prints the
# bench line's verdict and the REPLAY-DIFF diagnostics (which op, which tensor, which variables' gradients).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
export IGAN_GRAPH_CHECK_VERBOSE=1
run() {   # $1 label, $2 runs, rest: env assignments
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 600 python bench.py --gpus 8 --one-gpu --backend gloo --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 > $O/out_${label}_$i.txt 2> $O/err_${label}_$i.txt
    python - "$label" "$i" "$O/out_${label}_$i.txt" <<'PY'
import sys, json
label, i, path = sys.argv[1:4]
lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
if not lines:
    print(label, 'run', i, 'NO RESULT'); sys.exit(0)
d = json.loads(lines[-1]); g = d['hip_graphs']
print(label, 'run', i, 'value %.1f' % d['value'], 'faithful', g['faithful'], g['checks'])
PY
    grep -h "REPLAY-DIFF\|WARNING: hipGraph" $O/out_${label}_$i.txt $O/err_${label}_$i.txt | cut -c1-700 | sort | uniq -c | sort -rn | head -12
  done
}
run default 4 IGAN_X=1
run eager_twice 3 IGAN_GRAPH_CHECK_EAGER_TWICE=1
run wgrad_fp32 3 IGAN_WGRAD_PLANES=0
run bf16 2 IGAN_CONV_PLANES=1
run slice_outer 2 IGAN_F16_TAP_OUTER=0
run fp32 2 IGAN_CONV_PLANES=0
