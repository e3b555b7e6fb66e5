#!/bin/bash
# Round-6 GPU sessions, one parameterised runner (replaces the one-off r5_*.sh recipes): `gpurun --timeout T -- 'bash tools/r6.sh <stage> [args]'`.
# Every stage writes under gpurun_out/r6/<stage>/ ; what is judged is copied into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
stage=$1; shift
O=gpurun_out/r6/$stage; mkdir -p "$O"
run() { # label, timeout, command...
  local label=$1 t=$2; shift 2
  echo "=== $label: $*" | tee -a "$O/log.txt"
  timeout "$t" "$@" > "$O/$label.txt" 2>&1; local rc=$?
  echo "rc $rc" >> "$O/$label.txt"; tail -n 3 "$O/$label.txt" | tee -a "$O/log.txt"
}
case $stage in
second_order)   # the regulariser steps per arithmetic form, from one state
  run test_reg_forms 1500 python -m pytest tests/test_gpu_reg_forms.py -m gpu -x -q -s
  run cfg2_init 900 python tools/reg_forms.py --res 32 --state init --pl-fracs 0,0.9,0.98 --variants "0;1;2;2:1024" --out "$O/cfg2_init.json"
  run cfg2_loop 1500 python tools/reg_forms.py --res 32 --state loop:4,8,16 --variants "0;1;2;2:1024" --out "$O/cfg2_loop.json"
  run bench_init 1200 python tools/reg_forms.py --res 128 --state init --pl-fracs 0.5,0.98 --variants "0;1;2" --ops G_reg --out "$O/bench_init.json"
  for f in 0 1 2; do IGAN_CONV_PLANES=$f run audit_greg_form$f 900 python tools/conv_audit.py --op G_reg; done
  ;;
second_order2)  # the per-call audit of both second-order ops, the loop states with a second fp32 implementation beside them, the row thresholds on the clock
  for op in G_reg D_reg; do for f in 0 1 2; do IGAN_CONV_PLANES=$f run audit_${op}_form$f 900 python tools/conv_audit.py --op $op; done; done
  run cfg2_loop_cpu32 1500 python tools/reg_forms.py --res 32 --state loop:4,16 --variants "0;2;cpu32" --out "$O/cfg2_loop_cpu32.json"
  bash tools/r6.sh ab 2 - "IGAN_PLANES_MIN_ROWS=1024 IGAN_WGRAD_PLANES_MIN_ROWS=1024"
  ;;
thin_cache)     # round 6: thin_out rewrite + constant filter images: parity subset, microbench old / new library, bench A/B, the co-residency probe over every small kernel family
  run tests 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_planes_variant.py tests/test_gpu_networks.py tests/test_gpu_fullsize.py -m gpu -x -q
  IGAN_LIB=inclusivegan_amd/csrc/libigan_hip_oldthin.so run thin_old 300 python tools/kernel_bench.py thin 6 50
  run thin_new 300 python tools/kernel_bench.py thin 6 50
  bash tools/r6.sh ab 2 "IGAN_LIB=inclusivegan_amd/csrc/libigan_hip_oldthin.so IGAN_FILTER_CACHE=0" "IGAN_FILTER_CACHE=0" "-"
  PROBE_VICTIMS=dense,thin,lpips,nn1,scale_dot,smallconv,stream run probe_none 300 python tools/coresidency_probe.py none 4 0 15
  PROBE_VICTIMS=dense,thin,lpips,nn1,scale_dot,smallconv,stream run probe_fwd2 300 python tools/coresidency_probe.py fwd2 4 4 25
  PROBE_VICTIMS=dense,thin,lpips,nn1,scale_dot,smallconv,stream run probe_wgrad2 300 python tools/coresidency_probe.py wgrad2 4 4 25
  ;;
glue)           # torch glue of the path-length step by source line, with and without SumsqTapsFn; parity of the step; bench A/B
  IGAN_SUMSQ_FN=0 run glue_G_reg_composite 600 python tools/glue_attrib.py G_reg 60
  run glue_G_reg 600 python tools/glue_attrib.py G_reg 60
  run tests 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_reg_forms.py "tests/test_gpu_loop_parity.py::test_graphed_training_loop_every_op_small_width" "tests/test_gpu_loop_parity.py::test_graph_replay_equals_eager_bitwise" -m gpu -x -q -rP
  bash tools/r6.sh ab 2 "IGAN_SUMSQ_FN=0" "-"
  ;;
wgrad_walk)     # the weight gradient's walking addresses: bit-identity with the previous library, per-layer times alternating, parity, bench A/B
  run digest_ab 1500 bash tools/planes_sched_ab.sh inclusivegan_amd/csrc/libigan_hip_oldwgrad.so 0.3
  run tests 1500 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_f16_dynamic_range.py -m gpu -x -q
  bash tools/r6.sh ab 3 "IGAN_LIB=inclusivegan_amd/csrc/libigan_hip_oldwgrad.so" "-" "IGAN_LIB=inclusivegan_amd/csrc/libigan_hip_noslp.so"
  ;;
dstep)          # the first-order D step of early iterations of config 2 on ONE state under forms / thresholds / a second fp32 implementation (the all-ops run stops there at 1024 rows)
  run cfg2_D_loop 2400 python tools/reg_forms.py --res 32 --state loop:1,2,5 --loop-op D --variants "0;1;2;2:2048;cpu32" --out "$O/cfg2_D_loop.json"
  ;;
dstep_audit)    # the state before the D step of iteration 2 of config 2 (where every form sits 7e-3 from fp64 and PyTorch's CPU fp32 does not): every conv / dense call of that step against fp64
  run capture 1500 python tools/reg_forms.py --res 32 --state loop:1 --loop-op D --variants "2" --keep-state /tmp/dstep_state
  run audit_D_loss 900 python tools/conv_audit.py --op D_loss --state /tmp/dstep_state/state_0.npz --min-k 256 --samples 2048
  run kink_flips 900 python tools/kink_flips.py /tmp/dstep_state/state_0.npz
  ;;
audit)          # per-call audit of one op under the three forms: bash tools/r6.sh audit D_reg [extra args]
  op=${1:-G_reg}; shift
  for f in 0 1 2; do IGAN_CONV_PLANES=$f run audit_${op}_form$f 900 python tools/conv_audit.py --op "$op" "$@"; done
  ;;
ab)             # alternating bench A/B of environment settings on one box: bash tools/r6.sh ab <rounds> "<env A>" "<env B>" ...  ("-" = no setting)
  rounds=$1; shift
  for i in $(seq 1 "$rounds"); do for e in "$@"; do
    [ "$e" = "-" ] && ee="IGAN_NOOP=1" || ee="$e"
    env $ee timeout 900 python bench.py --data-size 1152 --no-cpu-baseline --no-roofline --no-variant-line --op-times 2>>"$O/err.txt" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d.get('op_ms'), d['hip_graphs']['faithful'])" | tee -a "$O/ab.txt"
  done; done
  ;;
evidence)       # the round's evidence set on the FINAL kernel sources: counter passes first (so that the bench line of the set carries roofline.traffic from passes
                # on these very sources), then bench line / kernel tables / per-layer tables / microbenchmarks (tools/collect_pmc.sh, tools/collect_profiles.sh)
  bash tools/collect_pmc.sh r06 > "$O/collect_pmc.log" 2>&1
  python tools/pmc_to_json.py r06 conv_fwd_planes_w4_kernel > "$O/pmc_to_json.log" 2>&1
  mkdir -p gpurun_out/prof_r06/pmc_json; cp profiles/r06_pmc* gpurun_out/prof_r06/pmc_json/ 2>/dev/null
  bash tools/collect_profiles.sh r06 > "$O/collect_profiles.log" 2>&1
  python tools/kernel_bench.py thin 6 50 >> gpurun_out/prof_r06/kernel_bench.txt 2>/dev/null
  tail -3 "$O/collect_pmc.log" "$O/pmc_to_json.log"
  ;;
driver)         # the driver's own command, twice in a row
  for i in 1 2; do run driver$i 1500 python bench.py --gpus 1 --steps 20 --warmup 5; done
  ;;
suite)          # the GPU suite with the tests' own prints kept (-rP)
  run gpu_tests 3000 python -m pytest tests -m gpu -q -rP --durations=15
  run smoke 600 python __graft_entry__.py smoke
  ;;
bench)          # the driver's command; extra args are passed on
  run bench 2400 python bench.py "$@"
  ;;
cmd)            # anything else: bash tools/r6.sh cmd <label> <timeout> <command...>
  run "$@"
  ;;
*) echo "unknown stage $stage"; exit 2;;
esac
