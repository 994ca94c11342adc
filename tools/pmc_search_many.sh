#!/bin/bash
# PMC passes over the many-query search (tools/search_many_probe.py, 64 queries): the filter pass (or the f32 scan) and the rescoring pass: pipe / LDS / L2 counters
set -u
N=${N:-1000000}
mkdir -p gpurun_out/pmcsm
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcsm/$name -- python tools/search_many_probe.py $N 64 10 10 > gpurun_out/pmcsm/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run f FETCH_SIZE
run w WRITE_SIZE
run d GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum
if python tools/pmc_summary.py gpurun_out/pmcsm > gpurun_out/pmcsm/summary_all.txt && grep -A28 "cosine_filter_bf16_kernel<12, 0\|cosine_rescore\|cosine_scan_mfma_kernel<1, 1" gpurun_out/pmcsm/summary_all.txt | tee gpurun_out/pmcsm/summary.txt | grep -q .; then
  find gpurun_out/pmcsm -name "*.csv" -delete
fi
