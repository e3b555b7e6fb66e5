#!/bin/bash
# SQ counter passes over one layer of tools/conv_layers.py: tools/pmc_layer.sh "<layer filter>" <kernel substring> [outfile]
R=$PWD; L=$1; K=$2; OUT=$(realpath -m ${3:-/dev/stdout})
export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift
  rm -rf /tmp/pl_$name
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pl_$name -- python3 $R/tools/conv_layers.py 0.05 "$L" > /tmp/pl_$name.log 2>&1 < /dev/null
  python3 $R/tools/pmc_summary.py /tmp/pl_$name $K < /dev/null
}
{ run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA
  run sq3 SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
} > $OUT 2>&1
