"""Shader cycles per phase of the decode attention's register path and of the one-row GEMV (tuning build only), over a Whisper or LLM decode.
usage: KJARNI_FFI_LIB=.../libkjarni_ffi_tuning.so python tools/attention_stamps.py whisper|llm
(kjarni_hip_attention_stamps(buf, 1) zeroes the counters and switches the stamps on, (buf, 0) reads them and switches them off:
while they are off the tuning build decodes at the shipped build's speed)"""
import ctypes as C, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kjarni_amd
from kjarni_amd import _ffi
from tests import synth

L = _ffi.lib()
buf = (C.c_uint64 * 16)()
which = sys.argv[1]
with tempfile.TemporaryDirectory() as tmp:
    if which == "whisper":
        synth.whisper_model(tmp, seed=0, base=True)
        wm = kjarni_amd.HipWhisper(tmp)
        wm.encode_audio(synth.synthetic_audio(30.0, seed=1), fetch=False)
        wm.greedy([50258, 50259, 50359, 50363], False, 8)
        L.kjarni_hip_attention_stamps(buf, 1)
        wm.greedy([50258, 50259, 50359, 50363], False, 448)
    else:
        synth.llm_model(tmp, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=4096, eos_token_id=[])
        dec = kjarni_amd.HipDecoder(tmp, max_context=2048)
        prompt = np.random.default_rng(0).integers(1000, 100000, 128).tolist()
        dec.generate(prompt, 8)
        L.kjarni_hip_attention_stamps(buf, 1)
        dec.generate(prompt, 256)
    L.kjarni_hip_attention_stamps(buf, 0)
n = max(1, buf[4])
names = ["entry -> scores (loads + dots)", "-> block max", "-> exp, weighted V, block sum", "-> slab stored"]
print(f"{which}: decode attention, {buf[4]} workgroups; cycles per workgroup: " + "; ".join(f"{nm} {buf[i] / n:.0f}" for i, nm in enumerate(names)) + f"; total {sum(buf[i] for i in range(4)) / n:.0f}")
if buf[11]:
    m = buf[11]
    print(f"{which}: one-row LN GEMV ({m} waves sampled), cycles from the wave's entry: arguments arrived {buf[9] / m:.0f}; weight requests issued {buf[10] / m:.0f}; every request issued {buf[12] / m:.0f}; the input row arrived {buf[8] / m:.0f}")
if buf[7]:
    print(f"{which}: one-row GEMV ({buf[7]} waves sampled): entry -> dot reduced {buf[5] / buf[7]:.0f} cycles, -> stored {buf[6] / buf[7]:.0f}")
