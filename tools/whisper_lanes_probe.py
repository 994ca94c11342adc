import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import synth
import kjarni_amd
with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "whisper-base")
    synth.whisper_model(d, seed=0, base=True)
    tr = kjarni_amd.Transcriber(model_path=d, max_tokens=448)
    audio = np.concatenate([synth.synthetic_audio(30.0, seed=20 + i) for i in range(16)])
    for lanes in ("1", "2", "4", "8"):
        os.environ["KJARNI_HIP_WHISPER_LANES"] = lanes
        tr.transcribe_audio(audio, 16000)
        t0 = time.perf_counter(); tr.transcribe_audio(audio, 16000); dt = time.perf_counter() - t0
        steps = 449 * (16 // int(lanes))
        print(f"lanes {lanes}: {dt:.3f} s, {480/dt:.0f}x real time, {dt/steps*1e3:.3f} ms per step", flush=True)
