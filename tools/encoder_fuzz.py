"""Random (batch, padded length) parity sweep of the encoder forward against the oracle, ragged masks, across the three
projection routes (<= 64 tokens, 65 .. 8192, more): python tools/encoder_fuzz.py [cases] [seed]."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from tests import synth
import kjarni_amd
from oracle import oracle as O


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "m")
        cfg, t = synth.minilm_embedder(d, seed=5, num_hidden_layers=2)
        enc = kjarni_amd.HipEncoder(d, 0)
        orc = O.OracleModel(t, cfg)
        worst = 0.0
        for i in range(cases):
            seq = int(rng.integers(1, 160))
            tokens = int(rng.choice([40, 64, 65, 300, 2000, 8192, 8300, 20000]))
            b = max(1, tokens // seq)
            ids, mask = synth.synthetic_ids(b, seq, seed=100 + i, ragged=True)
            got = enc.embed(ids, mask)
            ref = orc.embed_batch(ids, mask)
            err = float(np.abs(got - ref).max())
            worst = max(worst, err)
            print(f"case {i}: batch {b} x seq {seq} = {b * seq} tokens: max abs err {err:.2e}", flush=True)
            if err >= 1e-4:
                sys.exit(1)
        print(f"{cases} cases ok, worst {worst:.2e}")


if __name__ == "__main__":
    main()
