#!/bin/bash
# PMC counter passes over the kernel micro-benchmark (separate rocprofv3 runs per counter set,
# as MI355X_MICROARCH.md prescribes; never combined with trace domains other than kernel-trace).
set -u
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+" | sort -u > gpurun_out/pmc/counters_available.txt
wc -l gpurun_out/pmc/counters_available.txt
run() { # name, counters...
  name=$1; shift
  KB_VARIANTS=${KB_VARIANTS:-0,2} KB_ITERS=2 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python tools/kernel_bench.py > gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"; find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -2
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run sq2 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA
run fetch FETCH_SIZE
run write WRITE_SIZE
python tools/pmc_summary.py gpurun_out/pmc | tee gpurun_out/pmc/summary.txt
