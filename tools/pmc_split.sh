#!/bin/bash
# PMC passes over the f32-on-bf16 projections (tools/split_probe.py): effective clock, MFMA pipe busy, LDS conflicts.
set -u
mkdir -p gpurun_out/pmcsp
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export KJARNI_HIP_F32_ON_BF16=1
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcsp/$name -- python tools/split_probe.py child 262144 3 > gpurun_out/pmcsp/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
python tools/pmc_summary.py gpurun_out/pmcsp durations | tee gpurun_out/pmcsp/summary.txt
