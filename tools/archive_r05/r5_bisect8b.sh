#!/bin/bash
# second bisect of the eight-process nondeterminism of the fp16 form: which layer class?  (two EAGER executions of each op are compared: IGAN_GRAPH_CHECK_EAGER_TWICE=1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
make -C inclusivegan_amd/csrc variant VARIANT=diag DEFS=-DIGAN_DIAGNOSTIC > $O/build.txt 2>&1
export IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_diag.so IGAN_GRAPH_CHECK_EAGER_TWICE=1 IGAN_GRAPH_CHECK_VERBOSE=1 IGAN_WGRAD_PLANES=0
run() {
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 600 python bench.py --gpus 8 --one-gpu --backend gloo --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 > $O/out_${label}_$i.txt 2> $O/err_${label}_$i.txt
    python - "$label" "$i" "$O/out_${label}_$i.txt" <<'PY'
import sys, json
label, i, path = sys.argv[1:4]
lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
if not lines:
    print(label, 'run', i, 'NO RESULT'); sys.exit(0)
d = json.loads(lines[-1]); g = d['hip_graphs']
print(label, 'run', i, 'faithful', g['faithful'])
PY
    grep -h "WARNING: hipGraph" $O/out_${label}_$i.txt $O/err_${label}_$i.txt | sed 's/.*training op \(.[A-Za-z_]*.\).*max |diff|: \(.*\)); running.*/\1 \2/' | cut -c1-120 | sort | uniq -c | sort -rn | head -6
  done
}
run all 2 IGAN_X=1
run cin128 2 IGAN_PLANES_ONLY_CIN=128
run cin256 2 IGAN_PLANES_ONLY_CIN=256
run cin512 2 IGAN_PLANES_ONLY_CIN=512
run noact 2 IGAN_PLANES_NO_ACT=1
run onlyact 2 IGAN_PLANES_NO_ACT=2
run noscale 2 IGAN_PLANES_NO_SCALE=1
run onlyscale 2 IGAN_PLANES_NO_SCALE=2
run plainkind 2 IGAN_PLANES_KIND=1
run stride2 2 IGAN_PLANES_KIND=2
run up2 2 IGAN_PLANES_KIND=3
run fwdonly 2 IGAN_PLANES_WT=1
run dgradonly 2 IGAN_PLANES_WT=2
