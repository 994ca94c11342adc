#!/bin/bash
# Crossover of the few-rows kernel (row groups over the grid) against the 64 x 64-tile route: whole forwards at several call
# sizes with the bound moved (tuning build, GEMM_VARIANT = 1000 + rows).
set -u
export KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so
for shape in "1 128" "2 128" "4 128" "8 128" "16 128" "32 128"; do
  set -- $shape
  rows=$(( $1 * $2 ))
  for x in 64 $rows; do
    echo -n "few-rows up to $x rows: "
    GEMM_VARIANT=$((1000 + x)) python tools/mid_probe.py $1 $2 1000 | tail -1
  done
done
