#!/bin/bash
# Collects the evidence set of profiles/ on a GPU box (run through gpurun from the repo root):
#   bench line (default flags), rocprofv3 kernel stats of the same command, per-layer conv microbenchmark,
#   per-shape conv accounting of one iteration, and the PMC passes (MFMA busy, HBM bytes) on the headline shape.
# Everything lands in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_n1.json
python tools/conv_bench.py 6 10 > $OUT/conv_layers_microbench.txt 2>/dev/null
python tools/conv_shapes.py > $OUT/conv_shapes.txt 2>/dev/null
python tools/kernel_bench.py conv 6 20 > $OUT/kernel_bench.txt 2>/dev/null
python tools/kernel_bench.py upfirdn 6 20 >> $OUT/kernel_bench.txt 2>/dev/null
python tools/kernel_bench.py epilogue 6 20 >> $OUT/kernel_bench.txt 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -o bench -- python3 $OLDPWD/bench.py --no-cpu-baseline > $OUT/bench_profiled.log 2>&1
cp $(find /tmp/pb_$TAG -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/pmc1_$TAG -- python3 $OLDPWD/tools/kernel_bench.py conv 6 5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc2_$TAG -- python3 $OLDPWD/tools/kernel_bench.py conv 6 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc3_$TAG -- python3 $OLDPWD/tools/kernel_bench.py conv 6 5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc4_$TAG -- python3 $OLDPWD/tools/kernel_bench.py upfirdn 6 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_$TAG -- python3 $OLDPWD/tools/kernel_bench.py upfirdn 6 5 > /dev/null 2>&1
cd $OLDPWD
for i in 1 2 3; do python tools/pmc_summary.py /tmp/pmc${i}_$TAG conv_fwd_kernel; done > $OUT/pmc.txt
for i in 4 5; do python tools/pmc_summary.py /tmp/pmc${i}_$TAG upfirdn2d; done >> $OUT/pmc.txt
python tools/prof_summary.py $OUT/bench_kernel_stats.csv 45 > $OUT/bench_kernel_stats_summary.txt
ls -la $OUT
