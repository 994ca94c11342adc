#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_ffi.py tests/test_gpu_families.py -m gpu -q 2>&1 | tail -5
./tools/lab/mfma_lab 2>&1 | grep -E "K=  384 N= 1536|K= 1536 N=  384" | awk 'NR%2==0' | tee gpurun_out/lab_ref.log
KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so python tools/gemm_probe.py 2>&1 | tee gpurun_out/gemm_probe2.log
python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
