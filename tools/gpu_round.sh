#!/bin/bash
# One GPU-box round: parity tests, the bench line (embed + rerank) with live roofline and the CPU leg.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time python -m pytest tests -m gpu -q 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 3 --warmup 1 ${BENCH_ARGS:-} 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
python bench.py --workload rerank --steps 2 --warmup 1 2>gpurun_out/bench_rerank.err | tee gpurun_out/bench_rerank.json
tail -3 gpurun_out/bench_rerank.err
