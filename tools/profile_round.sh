#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats of bench.py and of the cosine scan, PMC passes over
# the pipeline, then the plain bench line and the extra measurements (tools/bench_more.py).
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof gpurun_out/prof_scan gpurun_out/pmcb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-rerank-leg --no-parity-check > gpurun_out/bench_prof.json 2>gpurun_out/prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scan -- python tools/bench_more.py scan > gpurun_out/scan_prof.jsonl 2>gpurun_out/prof_scan.err
rm -rf gpurun_out/prof_llm
KJARNI_BENCH_NO_CPU=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_llm -- python tools/bench_more.py llm > gpurun_out/llm_prof.jsonl 2>gpurun_out/prof_llm.err
find gpurun_out/prof gpurun_out/prof_scan gpurun_out/prof_llm -name "*kernel_trace.csv" -delete   # gpurun_out travels back: keep the summaries only
find gpurun_out/prof gpurun_out/prof_scan gpurun_out/prof_llm -name "*kernel_stats*"
bash tools/pmc_bench.sh > gpurun_out/pmcb.log 2>&1
tail -3 gpurun_out/pmcb.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tee gpurun_out/bench.json
python tools/bench_more.py 2>gpurun_out/bench_more.err | tee gpurun_out/bench_more.jsonl
