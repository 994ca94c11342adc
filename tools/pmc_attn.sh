#!/bin/bash
# PMC passes over the attention kernel alone (tools/attn_probe.py).
set -u
ATTN_ARGS=${ATTN_ARGS:-}
mkdir -p gpurun_out/pmca
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmca/$name -- python tools/attn_probe.py $ATTN_ARGS > gpurun_out/pmca/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS SQ_LEVEL_WAVES
find gpurun_out/pmca -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py gpurun_out/pmca | tee gpurun_out/pmca/summary.txt
python tools/attn_probe.py $ATTN_ARGS
