#!/bin/bash
# Counter passes of the evidence set (run through gpurun from the repo root, after tools/collect_profiles.sh):
#   headline modulated conv: issue / stall / LDS / instruction mix, HBM-side traffic and L2 hit rate with the XCD-aware block
#   order off and on (tools/pmc_conv.sh); upfirdn2d traffic at its three call sites; the conv family over the eager device work
#   of one G step and one D step (per-instantiation average traffic: what bench.py reports as roofline.traffic).
# Each pass is its own rocprofv3 run with --pmc + --kernel-trace only.  Everything lands in gpurun_out/prof_<tag>/pmc/.
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG/pmc
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
tools/pmc_conv.sh $OUT 6 > /dev/null 2>&1
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
