"""Soak check of the decode loop on the Llama-1B shape: long generations across the attention-split boundaries, twice with
the same inputs (greedy must repeat token for token), and sampled runs with a fixed seed (must repeat as well)."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402
import kjarni_amd  # noqa: E402
from kjarni_amd.chat import Chat, GenerationConfig  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "m")
    synth.llm_model(d, synth.LLAMA_1B, seed=0, store_bf16=True, max_position_embeddings=8192, bos_token_id=700, eos_token_id=[701])
    shutil.copy(os.path.join(ROOT, "tests", "golden", "bpe_llama3_tokenizer.json"), os.path.join(d, "tokenizer.json"))
    dec = kjarni_amd.HipDecoder(d, max_context=8192)
    prompt = np.random.default_rng(0).integers(1000, 100000, 700).tolist()
    t0 = time.perf_counter()
    a = dec.generate(prompt, 6000)
    dt = time.perf_counter() - t0
    b = dec.generate(prompt, 6000)
    assert a == b and len(a) == 6000, (len(a), len(b))
    print(f"greedy 6000 tokens after a 700-token prompt: {len(a) / dt:.0f} tokens/s incl. prefill, repeatable", flush=True)
    c = dec.generate(prompt, 300, repetition_penalty=1.2, no_repeat_ngram=3)   # host-processor path on the replayed graph
    e = dec.generate(prompt, 300, repetition_penalty=1.2, no_repeat_ngram=3)
    assert c == e and len(c) == 300
    del dec
    chat = Chat("llama3.2-1b-instruct", model_path=d)
    runs = []
    for _ in range(2):
        chat.seed(1234)
        pieces = []
        chat.stream("Tell me about Iceland. " * 20, lambda t: pieces.append(t) or True, GenerationConfig(max_new_tokens=1500, top_k=50, temperature=0.9))
        runs.append(pieces)
    assert runs[0] == runs[1] and len(runs[0]) == 1500
    convo = chat.conversation()
    for turn in range(6):
        r = convo.send(f"turn {turn}: say more", GenerationConfig(do_sample=False, max_new_tokens=64))
        assert isinstance(r, str)
    assert len(convo) == 12
    print("sampled 1500 tokens x2 with a fixed seed: identical; 6-turn conversation ok", flush=True)
