#!/bin/bash
# PMC passes on the headline modulated conv (tools/kernel_bench.py conv B reps): issue / stall / LDS / instruction-mix counters
# and HBM-side traffic with the XCD-aware block order off and on.  Run on a GPU box from the repo root:
#   tools/pmc_conv.sh <outdir> [batch]
# Counter passes are separate runs with --kernel-trace only (no other trace domains).
OUT=$(realpath -m ${1:?outdir}); B=${2:-6}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
run() {  # name, counters..., env prefix handled by caller
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $R/tools/kernel_bench.py conv $B 5 > /tmp/pmc_$name.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$name conv_fwd rows_f16 filter_amax filter_planes to_planes > $OUT/pmc_$name.txt 2>&1      # the tile kernel and the image kernels of the same call
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16
run sq3 SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_LDS
export IGAN_XCD_REMAP=0
run fetch_plain FETCH_SIZE
run write_plain WRITE_SIZE
export IGAN_XCD_REMAP=1
run fetch_xcd FETCH_SIZE
run write_xcd WRITE_SIZE
run l2_xcd TCC_HIT_sum TCC_MISS_sum
export IGAN_XCD_REMAP=0
run l2_plain TCC_HIT_sum TCC_MISS_sum
cd $R
tail -n +1 $OUT/pmc_*.txt
