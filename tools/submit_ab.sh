#!/bin/bash
# Round 4, host-submission experiments on one box (output: gpurun_out/submit_ab.txt):
#   1. tools/piece_shape_probe: 32x32x16 vs paired-piece 16x16x32 step of the piece kernels
#   2. bench A/B of HIP-runtime settings that bound how far the host may run ahead of the device (packet capture OFF, the product setting)
#   3. the graph-replay probe with packet capture forced ON under kernarg / flush settings (is the round-3 fault a kernarg or a fence problem?)
#   4. the bench with packet capture ON (the unfaithful op falls back to eager execution): what a faithful fast replay would be worth
OUT=gpurun_out/submit_ab.txt
mkdir -p gpurun_out
: > $OUT
echo "## 1. piece shape probe" >> $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/piece_shape_probe.hip -o /tmp/piece_shape_probe >> $OUT 2>&1 && timeout 120 /tmp/piece_shape_probe 1.0 >> $OUT 2>&1
B="python bench.py --data-size 1152 --no-cpu-baseline --no-roofline --no-variant-line --steps 48 --warmup 8 --op-times"
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'dev', d.get('op_ms'), 'host', d.get('op_host_ms'), 'faithful', d['hip_graphs'].get('faithful'))
except Exception as e: print('$1', 'FAILED', repr(e))"; }
echo "## 2. bench, packet capture off" >> $OUT
for cfg in "X=0" "HSA_KERNARG_POOL_SIZE=33554432" "ROC_SIGNAL_POOL_SIZE=8192" "DEBUG_CLR_MAX_BATCH_SIZE=100000" "X=1"; do
  env $cfg timeout 600 $B 2>/dev/null | line "$cfg" >> $OUT
done
echo "## 3. replay probe, packet capture ON" >> $OUT
for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "GPU_FLUSH_ON_EXECUTION=1" "DEBUG_CLR_BLIT_KERNARG_OPT=0" "ROC_SKIP_KERNEL_ARG_COPY=0" "HIP_FORCE_DEV_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0"; do
  echo "# $cfg" >> $OUT
  env DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 IGAN_GRAPH_VALIDATE=0 $cfg timeout 300 python tools/graph_replay_probe.py 2>&1 | grep -E "replay|eager" | head -12 >> $OUT
done
echo "## 4. bench, packet capture ON (unfaithful ops run eagerly)" >> $OUT
env DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 600 $B 2>gpurun_out/submit_ab_pc1.err | line "PACKET_CAPTURE=1" >> $OUT
grep -i "faithful\|eager\|disagree" gpurun_out/submit_ab_pc1.err | head -10 >> $OUT
cat $OUT
