#!/bin/bash
# PMC passes over the real pipeline (bench.py, 1 warm-up + 1 timed step).
set -u
mkdir -p gpurun_out/pmcb
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcb/$name -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-profile --no-rerank-leg --no-parity-check --no-scan --no-sensors --no-models --sentences 16384 > gpurun_out/pmcb/$name.log 2>&1
  echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run b GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run f FETCH_SIZE
run w WRITE_SIZE
python tools/pmc_summary.py gpurun_out/pmcb durations | tee gpurun_out/pmcb/summary.txt
