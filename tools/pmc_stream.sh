#!/bin/bash
# PMC passes over the K-stream tile kernel (tools/stream_probe.py): the product (DIAG 0) beside its knock-outs (no stores = 2, no epilogue = 1)
set -u
mkdir -p gpurun_out/pmcst
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcst/$name -- python tools/stream_probe.py 262144 0 52 9 > gpurun_out/pmcst/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
run d GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum
python tools/pmc_summary.py gpurun_out/pmcst > gpurun_out/pmcst/summary_all.txt 2>&1
grep -B1 -A30 "gemm_nt_f32_stream" gpurun_out/pmcst/summary_all.txt > gpurun_out/pmcst/summary.txt
find gpurun_out/pmcst -name "*.csv" -size +2M -delete
