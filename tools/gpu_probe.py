"""GPU-box probe: torch (its bundled HIP runtime) + libkjarni_ffi.so in one process."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kjarni_amd
from tests import synth

print("torch", torch.__version__, "cuda", torch.cuda.is_available(), "devices", kjarni_amd.device_count())
print(torch.cuda.get_device_name(0))
with tempfile.TemporaryDirectory() as tmp:
    cfg, t = synth.minilm_embedder(tmp, seed=0)
    enc = kjarni_amd.HipEncoder(tmp, 0)
    B, S = 4096, 128
    ids, mask = synth.synthetic_ids(B, S, seed=0)
    host = enc.embed(ids[:64], mask[:64])
    dev = torch.device("cuda:0")
    ids_t = torch.from_numpy(ids.view(np.int32)).to(dev)
    mask_t = torch.from_numpy(mask.view(np.int32)).to(dev)
    out_t = torch.empty((B, 384), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    enc.embed_dev(ids_t.data_ptr(), mask_t.data_ptr(), B, S, out_t.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    got = out_t[:64].cpu().numpy()
    print("torch-pointer path vs host path max diff:", np.abs(got - host).max())
    for chunk in (8192, 16384, 32768, 65536):
        enc.set_chunk_tokens(chunk)
        enc.embed_dev(ids_t.data_ptr(), mask_t.data_ptr(), B, S, out_t.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            enc.embed_dev(ids_t.data_ptr(), mask_t.data_ptr(), B, S, out_t.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        print(f"chunk_tokens={chunk}: {B/dt:.0f} sentences/s ({dt*1e3:.1f} ms per {B})")
