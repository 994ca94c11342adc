"""Encoder calls of a given size in a loop (for rocprofv3): python tools/mid_probe.py [batch] [seq] [reps]."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (before the HIP library: it brings its own runtime)
from tests import synth  # noqa: E402
import kjarni_amd  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seq = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
if os.environ.get("GEMM_VARIANT"):  # tuning build only (KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so)
    from kjarni_amd import ops
    ops.set_gemm_variant(int(os.environ["GEMM_VARIANT"]))
if os.environ.get("ATTN_VARIANT"):
    from kjarni_amd import ops
    ops.set_attention_variant(int(os.environ["ATTN_VARIANT"]))
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "m")
    if os.environ.get("MODEL") == "base":   # BERT-base shape (768 x 12 heads of 64, inner 3072), 6 layers
        synth.minilm_embedder(d, seed=0, hidden_size=768, num_attention_heads=12, intermediate_size=3072)
    else:
        synth.minilm_embedder(d, seed=0)
    enc = kjarni_amd.HipEncoder(d, 0)
    ids, mask = synth.synthetic_ids(b, seq, seed=1)
    enc.embed(ids, mask)
    t0 = time.perf_counter()
    for _ in range(reps):
        enc.embed(ids, mask)
    dt = (time.perf_counter() - t0) / reps
    print(f"batch {b} x {seq}: {dt * 1e3:.3f} ms per call = {b / dt:.0f} sentences/s", flush=True)
