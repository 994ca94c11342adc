#!/bin/bash
# PMC passes over encoder calls of 32 x 128 tokens (the mid-size GEMM route); pass d (cache counters) is slow: "full" only
set -u
mkdir -p gpurun_out/pmcm
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcm/$name -- python tools/mid_probe.py 32 128 10 > gpurun_out/pmcm/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES
if [ "${1:-}" = full ]; then
run d GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
fi
# (round 4: the mid-size projections are gemm_nt_f32_flex<RA, CB, VSLICES>; the raw CSVs go only once the summary exists)
if python tools/pmc_summary.py gpurun_out/pmcm > gpurun_out/pmcm/summary_all.txt && grep -A32 -E "gemm_nt_f32_flex|gemm_nt_f32_mid<" gpurun_out/pmcm/summary_all.txt | tee gpurun_out/pmcm/summary.txt | grep -q .; then
  find gpurun_out/pmcm -name "*.csv" -delete
fi
