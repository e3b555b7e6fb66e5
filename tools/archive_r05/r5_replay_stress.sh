#!/bin/bash
# the eight-ranks-on-one-GPU replay stress (tests/test_gpu_dist.py::test_eight_rank_replay_stress) on the product library and on a
# build whose weight-gradient kernel issues its LDS-DMA through the compiler builtin (-DIGAN_WGRAD_ASM_DMA=0: the build that was unfaithful in round 4).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
[ -f inclusivegan_amd/csrc/libigan_hip_builtindma.so ] || make -C inclusivegan_amd/csrc variant VARIANT=builtindma DEFS=-DIGAN_WGRAD_ASM_DMA=0 > $O/build.txt 2>&1      # built here when the snapshot did not bring it
run() {   # $1 label, $2 lib or ""
  for i in 1 2 3; do
    if [ -n "$2" ]; then export IGAN_LIB=$2; else unset IGAN_LIB; fi
    timeout 1500 python bench.py --gpus 8 --one-gpu --backend gloo --steps 17 --warmup 1 --no-roofline --no-cpu-baseline --data-size 1000 --num-samples-factor 1 --revalidate-every 1 2> $O/err_$1_$i.txt | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['hip_graphs']; print('$1 run $i:', d['value'], 'faithful', g['faithful'], 'checks', len(g['checks']), 'unfaithful', [c for c in g['checks'] if not c['faithful']][:4])"
    grep -c "does not reproduce" $O/err_$1_$i.txt
  done
}
run product "" 2>&1 | tee $O/stress.txt
run builtindma $PWD/inclusivegan_amd/csrc/libigan_hip_builtindma.so 2>&1 | tee -a $O/stress.txt
