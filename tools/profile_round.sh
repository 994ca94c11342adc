#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats of bench.py + PMC passes over the pipeline.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof gpurun_out/pmcb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2>gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_stats*"
bash tools/pmc_bench.sh > gpurun_out/pmcb.log 2>&1
tail -3 gpurun_out/pmcb.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tee gpurun_out/bench.json
