#!/bin/bash
# Collects the evidence set of profiles/ on a GPU box (run through gpurun from the repo root):
#   bench line (default flags), rocprofv3 kernel stats + steady-state kernel breakdown of the same command, the stamped per-shape
#   conv table, per-layer sustained microbenchmark, single-kernel microbenchmarks, full-size IMLE refresh.
# Everything lands in gpurun_out/prof_<tag>/; copy what should be judged into profiles/.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --conv-shapes $OUT/conv_shapes.txt 2>$OUT/bench.err | tail -1 > $OUT/bench_n1.json        # the driver's command: full-size data set, refresh included
python tools/conv_layers.py 0.2 > $OUT/conv_layers.txt 2>/dev/null
IGAN_CONV_PLANES=0 python tools/conv_layers.py 0.2 > $OUT/conv_layers_exact_fp32.txt 2>/dev/null       # the labelled lines' kernels
IGAN_CONV_PLANES=1 python tools/conv_layers.py 0.2 > $OUT/conv_layers_bf16_pieces.txt 2>/dev/null
python tools/kernel_bench.py upfirdn 6 40 > $OUT/kernel_bench.txt 2>/dev/null
python tools/kernel_bench.py epilogue 6 40 >> $OUT/kernel_bench.txt 2>/dev/null
python tools/conv_sustain.py 1.5 2 4 6 8 12 >> $OUT/kernel_bench.txt 2>/dev/null
python tools/conv_phases.py 2 4 6 12 > $OUT/conv_phases.txt 2>/dev/null
tools/mfma_rate 2 > $OUT/mfma_rate.txt 2>&1
python tools/refresh_fullsize.py > $OUT/refresh_fullsize.json 2>$OUT/refresh_fullsize.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf_$TAG -o rf -- python3 $R/tools/refresh_fullsize.py 30000 1 > $OUT/refresh_profiled.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/rf_$TAG -name "*kernel_stats.csv" | head -1) 14 > $OUT/refresh_kernel_stats_tenth.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -o bench -- python3 $R/bench.py --no-cpu-baseline --no-variant-line --data-size 1152 > $OUT/bench_profiled.log 2>&1      # small data set: the trace is about the training step, not the refresh
cp $(find /tmp/pb_$TAG -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
cd $R
python tools/prof_summary.py $OUT/bench_kernel_stats.csv 45 > $OUT/bench_kernel_stats_summary.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -o kt -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-variant-line --steps 24 --data-size 1152 > $OUT/bench_traced.log 2>&1
cd $R
python tools/gpu_idle.py $(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1) 0.3 12 > $OUT/bench_steady_state.txt 2>&1
ls -la $OUT
# counter passes: tools/collect_pmc.sh <tag> (its own gpurun call; they take as long as everything above)
