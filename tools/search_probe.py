"""kjarni_hip_cosine_search (one query, fused scan + selection) in a loop, for rocprofv3 --kernel-trace --stats:
python tools/search_probe.py [n_docs] [k] [reps] [dim]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kjarni_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 384
L = _ffi.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(2)
corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
q = torch.randn((1, dim), generator=g, device=dev, dtype=torch.float32)
ws = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(1, n, dim, k), dtype=torch.uint8, device=dev)
idx = torch.empty((1, k), dtype=torch.int64, device=dev)
sc = torch.empty((1, k), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), 1, corpus.data_ptr(), n, dim, 1, k, ws.data_ptr(), idx.data_ptr(),
                                                sc.data_ptr(), st))


for _ in range(20):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"n {n} dim {dim} k {k}: {dt * 1e6:.1f} us per search = {n * dim * 4 / dt / 1e9:.0f} GB/s", flush=True)
