#!/bin/bash
# f32-on-bf16 mode: from how many rows should the split kernel's 128 x 128 tiles replace the 64 x 64-tile route?  Whole forwards at
# several call sizes, mode off and on with the bound moved (tuning build, GEMM_VARIANT = 100000 + rows).
export KJARNI_FFI_LIB=kjarni_amd/lib/libkjarni_ffi_tuning.so
for shape in "8 128" "16 128" "32 128" "64 128"; do
  set -- $shape
  echo "== $1 x $2"
  KJARNI_HIP_F32_ON_BF16=0 python tools/mid_probe.py $1 $2 400 | tail -1 | sed "s/^/mode off:          /"
  for x in 1024 2048 4096 9000; do
    KJARNI_HIP_F32_ON_BF16=1 GEMM_VARIANT=$((100000 + x)) python tools/mid_probe.py $1 $2 400 | tail -1 | sed "s/^/mode on, from $x: /"
  done
done
