#!/bin/bash
# Crossover of the few-rows kernel against the 64 x 64-tile route on a 768-wide model (tuning build, GEMM_VARIANT = 1000 + rows)
set -u
export KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so MODEL=base
for shape in "1 64" "1 128" "2 96" "2 128" "3 128"; do
  set -- $shape
  rows=$(( $1 * $2 ))
  for x in 64 $rows; do
    GEMM_VARIANT=$((1000 + x)) python tools/mid_probe.py $1 $2 500 | tail -1 | sed "s/^/bound $x: /"
  done
done
