#!/bin/bash
# Counter passes of the evidence set (run through gpurun from the repo root, after tools/collect_profiles.sh):
#   headline modulated conv: issue / stall / LDS / instruction mix, HBM-side traffic and L2 hit rate with the XCD-aware block
#   order off and on (tools/pmc_conv.sh); upfirdn2d traffic at its three call sites; the conv family over the eager device work
#   of one G step and one D step (per-instantiation average traffic: what bench.py reports as roofline.traffic).
# Each pass is its own rocprofv3 run with --pmc + --kernel-trace only.  Everything lands in gpurun_out/prof_<tag>/pmc/.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG/pmc
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
# the kernel sources these passes were taken on: bench.py refuses a traffic figure whose recorded hashes differ from the tree's
python3 - > $OUT/kernel_source_sha16.json <<PY
import hashlib, json
print(json.dumps({f: hashlib.sha256(open('$R/inclusivegan_amd/csrc/' + f, 'rb').read()).hexdigest()[:16] for f in ('conv2d_mfma.hip', 'upfirdn2d.hip')}))
PY
tools/pmc_conv.sh $OUT 6 > /dev/null 2>&1
# the weight-gradient family on the three largest layers: SQ counters + HBM-side traffic
i=0
for L in "G 32 Conv1" "G 64 Conv1" "G 128 Conv1"; do
  i=$((i+1))
  tools/pmc_layer.sh "$L" conv_wgrad $OUT/pmc_wgrad_layer$i.txt
  tools/pmc_traffic.sh "$L" conv_wgrad > $OUT/pmc_wgrad_layer${i}_traffic.txt 2>&1
done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcu_${c}_$TAG -- python3 $R/tools/kernel_bench.py upfirdn 6 5 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcu_${c}_$TAG upfirdn2d > $OUT/pmc_upfirdn_$c.txt 2>&1
  for op in G_train D_train; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmco_${op}_${c}_$TAG -- python3 $R/tools/op_profile.py $op 2 > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py /tmp/pmco_${op}_${c}_$TAG conv_fwd conv_wgrad > $OUT/pmc_${op}_$c.txt 2>&1
  done
done
cd $R
tail -n +1 $OUT/pmc_*.txt
