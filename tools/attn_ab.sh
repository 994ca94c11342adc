#!/bin/bash
# Same-box A/B of the attention kernel between two builds of the library (KJARNI_FFI_LIB): alternating processes, headline shape
# (2 048 sentences x 128 tokens x 12 heads x 32) and the d = 64 shape.  usage: tools/attn_ab.sh libA.so libB.so
A=${1:-$PWD/kjarni_amd/lib/libkjarni_ffi_prev.so}; B=${2:-$PWD/kjarni_amd/lib/libkjarni_ffi.so}
for round in 1 2 3; do
  for lib in "$A" "$B"; do
    echo -n "$(basename $lib): "; KJARNI_FFI_LIB=$lib python tools/attn_probe.py 2048 128 12 32 | tr '\n' ' '; KJARNI_FFI_LIB=$lib python tools/attn_probe.py 1024 128 12 64
  done
done
