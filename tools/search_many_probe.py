"""kjarni_hip_cosine_search with many queries (selection inside the matrix-core scan) in a loop, for rocprofv3:
python tools/search_many_probe.py [n_docs] [nq] [k] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from kjarni_amd import _ffi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # (tuning build: kjarni_hip_set_cosine_variant)
dim = int(os.environ.get("DIM", "384"))
L = _ffi.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(2)
corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
ws = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, n, dim, k), dtype=torch.uint8, device=dev)
idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
if variant:
    from kjarni_amd import ops
    ops.set_cosine_variant(variant)


def run():
    _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, 1, k, ws.data_ptr(), idx.data_ptr(),
                                                sc.data_ptr(), st))


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"n {n} nq {nq} k {k}{' variant ' + str(variant) if variant else ''}: {dt * 1e3:.3f} ms per search", flush=True)
