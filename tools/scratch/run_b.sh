export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_refresh.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
python tools/refresh_fullsize.py 30000 1 2>/dev/null | tail -1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf -o rf -- python3 $R/tools/refresh_fullsize.py 30000 1 > $R/gpurun_out/refresh_prof.log 2>&1
cd $R
python tools/prof_summary.py $(find /tmp/rf -name "*kernel_stats.csv" | head -1) 12 > gpurun_out/refresh_kernel_stats.txt
cat gpurun_out/refresh_kernel_stats.txt
