#!/bin/bash
# One GPU-box round: parity tests, kernel micro-benchmarks, bench with live roofline.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
KB_VARIANTS=0 python tools/kernel_bench.py 2>&1 | tee gpurun_out/kernel_bench.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
