"""Prefill-only probe on the Llama-1B shape (for rocprofv3): python tools/prefill_probe.py [tokens] [repeats]."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402
import kjarni_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "m")
    synth.llm_model(d, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=8192, eos_token_id=[])
    dec = kjarni_amd.HipDecoder(d, max_context=4096)
    prompt = np.random.default_rng(0).integers(1000, 100000, n).tolist()
    dec.forward(prompt, fetch=False)
    for _ in range(reps):
        dec.reset()
        t0 = time.perf_counter()
        dec.forward(prompt, fetch=False)
        dt = time.perf_counter() - t0
        print(f"prefill {n} tokens: {dt * 1e3:.2f} ms = {n / dt:.0f} tokens/s", flush=True)
