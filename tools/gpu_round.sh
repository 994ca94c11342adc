#!/bin/bash
# One GPU-box round: parity tests, bench with live roofline, rocprofv3 kernel-trace stats.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/bench_prof.json 2>gpurun_out/prof.err
tail -3 gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_stats*" | head
