#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
./tools/lab/mfma_lab 2>&1 | tee gpurun_out/mfma_lab.log
( time python -m pytest tests -m gpu -q 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/pytest_gpu.log
