#!/bin/bash
# PMC passes over the 64-query cosine scan (cosine_scan_mfma_kernel): tools/bench_more.py scan
set -u
mkdir -p gpurun_out/pmcs
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcs/$name -- python tools/bench_more.py scan > gpurun_out/pmcs/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run f FETCH_SIZE
run w WRITE_SIZE
run d GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum
# (the raw CSVs go only once the summary exists: a failed summary keeps the run)
if python tools/pmc_summary.py gpurun_out/pmcs > gpurun_out/pmcs/summary_all.txt && grep -A28 "cosine_scan_mfma" gpurun_out/pmcs/summary_all.txt | tee gpurun_out/pmcs/summary.txt | grep -q .; then
  find gpurun_out/pmcs -name "*.csv" -delete
fi
