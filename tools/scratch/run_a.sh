export TMPDIR=/tmp
R=$PWD
python bench.py --data-size 1152 --no-cpu-baseline 2>gpurun_out/bench_a.err | tail -1 > gpurun_out/bench_a.json
cat gpurun_out/bench_a.json | cut -c1-700
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf -o rf -- python3 $R/tools/refresh_fullsize.py 30000 1 > $R/gpurun_out/refresh_prof.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/rf -name "*kernel_stats.csv" | head -1) 30 > gpurun_out/refresh_kernel_stats.txt
tail -2 gpurun_out/refresh_prof.log; cat gpurun_out/refresh_kernel_stats.txt
