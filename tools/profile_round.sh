#!/bin/bash
# Evidence for profiles/ (one GPU call): the headline in the DRIVER's form (--steps 20 --warmup 5) and a --steps 3 run from the
# same box, rocprofv3 kernel-trace stats of the same command, PMC passes over the pipeline (traffic for roofline.traffic),
# the in-process arrangement, and the other measurements (tools/bench_more.py).
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-scan --no-rerank-leg --no-models 2>>gpurun_out/bench.err | tee gpurun_out/bench_steps3.json
rm -rf gpurun_out/prof gpurun_out/pmcb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-scan --no-sensors --no-rerank-leg --no-parity-check --no-models > gpurun_out/bench_prof.json 2>gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_trace.csv" -delete   # gpurun_out travels back: keep the summaries only
find gpurun_out/prof -name "*kernel_stats*"
bash tools/pmc_bench.sh > gpurun_out/pmcb.log 2>&1
tail -3 gpurun_out/pmcb.log
python bench.py --gpus 1 --in-process --steps 5 --warmup 2 2>>gpurun_out/bench.err | grep "^{" | tee gpurun_out/bench_in_process.json
python tools/bench_more.py 2>gpurun_out/bench_more.err | tee gpurun_out/bench_more.jsonl
