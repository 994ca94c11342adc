#!/bin/bash
# PMC passes over gemm_flex.hip's tiles at 4 096 rows (tools/flex_knockout.py, tuning build); summary -> gpurun_out/pmc_flex_summary.txt
set -u
mkdir -p gpurun_out/pmcf
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcf/$name -- python tools/flex_knockout.py 20 ${ROWS:-4096} > gpurun_out/pmcf/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
run d GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
python tools/pmc_summary.py gpurun_out/pmcf > gpurun_out/pmc_flex_summary.txt && find gpurun_out/pmcf -name "*.csv" -delete
grep -A40 "flex<2, 12, false, 0>" gpurun_out/pmc_flex_summary.txt | head -60
