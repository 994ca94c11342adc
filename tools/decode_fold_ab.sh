#!/bin/bash
# Same-box A/B of the one-token decode steps' launch fusions (tuning build): embedding gather inside the first projection, final
# LayerNorm inside the vocabulary head (Whisper); embedding gather inside the first layer's norm + QKV + RoPE launch (LLM).
export KJARNI_FFI_LIB=$PWD/kjarni_amd/lib/libkjarni_ffi_tuning.so
for round in 1 2 3; do
  python tools/decode_probe.py whisper 2>/dev/null | tail -1
  KJARNI_HIP_WHISPER_NO_FOLD=1 python tools/decode_probe.py whisper 2>/dev/null | tail -1
  KJARNI_HIP_WHISPER_NO_FOLD=embed python tools/decode_probe.py whisper 2>/dev/null | tail -1
  KJARNI_HIP_WHISPER_NO_FOLD=head python tools/decode_probe.py whisper 2>/dev/null | tail -1
done
for round in 1 2 3; do
  python tools/decode_probe.py llm 2>/dev/null | tail -1
  KJARNI_HIP_LLM_NO_FOLD=1 python tools/decode_probe.py llm 2>/dev/null | tail -1
done
