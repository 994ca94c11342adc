#!/bin/bash
# Same-box A/B of the one-row GEMV's columns per wave / waves per workgroup (tuning build): ms per Whisper token + the stamps.
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
for w in 4 1; do for c in 1 2 4; do
  KJARNI_HIP_GEMV_COLS=$c KJARNI_HIP_GEMV_WAVES=$w python tools/decode_probe.py whisper 3 2>&1 | grep "ms/token"
  KJARNI_HIP_GEMV_COLS=$c KJARNI_HIP_GEMV_WAVES=$w python tools/attention_stamps.py whisper 2>&1 | grep "one-row"
done; done
