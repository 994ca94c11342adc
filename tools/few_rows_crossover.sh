#!/bin/bash
# Few-rows kernel against the per-call tile route around the 256-row bound (tuning build; GEMM_VARIANT = 1000 + rows): bash tools/few_rows_crossover.sh
set -u
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
for shape in "1 128" "1 160" "1 192" "2 112" "2 128" "5 48" "8 28"; do
  set -- $shape
  for x in 64 999; do
    echo -n "$1 x $2, few-rows up to $x rows: "
    GEMM_VARIANT=$((1000 + x)) python tools/mid_probe.py $1 $2 1500 2>/dev/null | grep ^batch
  done
done
