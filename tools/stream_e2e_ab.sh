#!/bin/bash
# Whole-step A/B on one box: the K-stream tile kernel (variant 0) against the per-tile-prologue kernel (53), alternating processes.
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
F="--steps 6 --warmup 3 --no-scan --no-sensors --no-cpu-baseline --no-extras --no-rerank-leg --no-parity-check --no-models"
for round in 1 2 3; do
  for v in 53 0; do
    python tools/bench_variant.py $v $F 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_one_step']
print('variant $v: %.1f sentences/s  %.2f ms/step  qkv %.1f fc1 %.1f fc2 %.1f out %.1f attn %.1f TFLOP/s  clock %.3f' % (d['value'], d['ms_per_step'], k['gemm_qkv']['tflops'], k['gemm_fc1']['tflops'], k['gemm_fc2']['tflops'], k['gemm_out_proj']['tflops'], k['attention']['tflops'], d['roofline'].get('clock_ghz_mean') or 0))"
  done
done
