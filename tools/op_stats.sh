mkdir -p gpurun_out/r02ac; export TMPDIR=/tmp; R=$PWD; cd /tmp
for op in G_reg D_reg G_train D_train; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ops_$op -o s -- python3 $R/tools/op_profile.py $op 6 > $R/gpurun_out/r02ac/$op.log 2>&1
  python3 $R/tools/prof_summary.py $(find /tmp/ops_$op -name "*kernel_stats.csv" | head -1) 60 > $R/gpurun_out/r02ac/$op.txt 2>&1
done
head -45 $R/gpurun_out/r02ac/G_reg.txt
