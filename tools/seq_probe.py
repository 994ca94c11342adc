"""Per-call times of the first 120 calls of one size after start-up (the clock ramp a sparse caller sees): python tools/seq_probe.py <sentences>."""
import os, sys, tempfile, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import torch
from tests import synth
import kjarni_amd
b = int(sys.argv[1]); seq = 128
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "m"); synth.minilm_embedder(d, seed=0)
    enc = kjarni_amd.HipEncoder(d, 0)
    ids, mask = synth.synthetic_ids(b, seq, seed=1)
    enc.embed(ids, mask)
    ts = []
    for _ in range(120):
        t0 = time.perf_counter(); enc.embed(ids, mask); ts.append((time.perf_counter() - t0) * 1e3)
    print(b, "first 40:", " ".join(f"{t:.2f}" for t in ts[:40]))
    print(b, "last 20:", " ".join(f"{t:.2f}" for t in ts[-20:]))
