"""Concurrency soak of one encoder handle: N host threads issue calls of mixed sizes (all three projection routes, workspaces
growing and being reused, two lanes, call combining) for a while.  Every result must equal the single-threaded result bit for
bit -- except small calls (<= 8 rows, <= 1 024 tokens) while combining is opted in (KJARNI_HIP_COMBINE=1): those share a forward
with whatever else is queued and are held to 1e-6 (by default, without it: bit for bit):
python tools/stress_encoder.py [threads] [seconds]."""
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import torch  # noqa: F401
from tests import synth
import kjarni_amd


def main():
    n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "m")
        synth.minilm_embedder(d, seed=5)
        enc = kjarni_amd.HipEncoder(d, 0)
        shapes = [(1, 12), (2, 30), (5, 64), (16, 40), (32, 128), (70, 128), (100, 90), (140, 128), (300, 128), (3, 500)]  # (140 x 128 ragged: ~10 000 kept tokens, always three parts)
        inputs = [synth.synthetic_ids(b, s, seed=50 + i, ragged=True) for i, (b, s) in enumerate(shapes)]
        refs = [enc.embed(i, m) for i, m in inputs]
        combining = os.environ.get("KJARNI_HIP_COMBINE", "0") not in ("", "0")   # (opt-in)
        small = [combining and b <= 8 and b * s <= 1024 for b, s in shapes]
        worst = [0.0]
        errors, calls = [], [0] * n_threads
        stop = time.time() + seconds

        def work(t):
            rng = np.random.default_rng(t)
            try:
                while time.time() < stop and not errors:
                    j = int(rng.integers(0, len(inputs)))
                    got = enc.embed(*inputs[j])
                    if small[j]:
                        d = float(np.abs(got - refs[j]).max())
                        worst[0] = max(worst[0], d)
                        if not d <= 1e-6:
                            errors.append(f"thread {t}: small call {shapes[j]} differs by {d:.3e} (> 1e-6)")
                    elif not np.array_equal(got, refs[j]):
                        errors.append(f"thread {t}: shape {shapes[j]} differs by {float(np.abs(got - refs[j]).max()):.3e}")
                    calls[t] += 1
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        print(f"{sum(calls)} calls from {n_threads} threads in {seconds:.0f} s: {('OK, large calls bit-identical' + (f', combined small calls within {worst[0]:.1e}' if combining else ', small calls too')) if not errors else errors[:3]}")
        sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
