"""Do two half-size calls on two streams beat one call?  N threads each embed B/N sentences x 128 tokens in a loop on one handle
(combining off; every call leases its own workspace + stream) against one thread with B sentences.
python tools/overlap_probe.py [B=32] [seq=128] [reps=300]"""
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from tests import synth  # noqa: E402
import kjarni_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seq = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "m")
    synth.minilm_embedder(d, seed=0)
    enc = kjarni_amd.HipEncoder(d, 0)
    enc.set_combining(False)
    enc.set_two_lanes(False)
    for n in (1, 2, 3, 4):
        b = B // n
        data = [synth.synthetic_ids(b, seq, seed=1 + i) for i in range(n)]
        for i in range(n):
            enc.embed(*data[i])

        def work(i):
            for _ in range(reps):
                enc.embed(*data[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = (time.perf_counter() - t0) / reps
        print(f"{n} thread(s) x {b} sentences x {seq}: {dt * 1e3:.3f} ms per round of {B} = {B / dt:.0f} sentences/s", flush=True)
