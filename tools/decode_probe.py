"""ms per greedy token of the two decode paths, for same-box A/Bs (tuning build env switches): best of 5 runs each.
usage: [KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so KJARNI_HIP_WHISPER_NO_FOLD=1|embed|head KJARNI_HIP_LLM_NO_FOLD=1] python tools/decode_probe.py whisper|llm [repetitions, default 5]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # (first: libkjarni_ffi.so then shares the HIP runtime torch ships, as in bench.py)
import kjarni_amd
from tests import synth

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("KJARNI_HIP_")) or "default"
with tempfile.TemporaryDirectory() as tmp:
    if which == "whisper":
        synth.whisper_model(tmp, seed=0, base=True)
        wm = kjarni_amd.HipWhisper(tmp)
        wm.encode_audio(synth.synthetic_audio(30.0, seed=1), fetch=False)
        prompt = [50258, 50259, 50359, 50363]
        wm.greedy(prompt, False, 8)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            ids = wm.greedy(prompt, False, 448)
            dt = (time.perf_counter() - t0) / len(ids)
            best = dt if best is None else min(best, dt)
        print(f"whisper [{tag}]: {best * 1e3:.4f} ms/token, first ids {ids[:6]}", flush=True)
    else:
        synth.llm_model(tmp, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=4096, eos_token_id=[])
        dec = kjarni_amd.HipDecoder(tmp, max_context=2048)
        prompt = np.random.default_rng(0).integers(1000, 100000, 128).tolist()
        dec.generate(prompt, 8)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            dec.reset()
            dec.forward(prompt, fetch=False)
            t1 = time.perf_counter()
            out = dec.generate(prompt, 256)
            dt = (time.perf_counter() - t0 - 2 * (t1 - t0)) / len(out)   # (generate() runs the prefill again)
            best = dt if best is None else min(best, dt)
        print(f"llm [{tag}]: {best * 1e3:.4f} ms/token, first ids {out[:6]}", flush=True)
