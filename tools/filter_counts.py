"""What the many-query search's passes leave behind (read out of the call's workspace after one search of 64 queries): overflow
word, final candidates per query, the bounds against the 10th best score, the filter waves' list lengths.
usage: python tools/filter_counts.py <n_docs>   (the workspace layout is launch_cosine_search's, cosine.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kjarni_amd import _ffi
n, nq, k, dim = int(sys.argv[1]), 64, 10, 384
L = _ffi.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(2)
corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
wsb = L.kjarni_hip_cosine_search_workspace_bytes(nq, n, dim, k)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
idx = torch.empty((nq, k), dtype=torch.int64, device=dev); sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
_ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, 1, k, ws.data_ptr(), idx.data_ptr(), sc.data_ptr(), st))
torch.cuda.synchronize()
p256 = lambda b: (b + 255) & ~255
off = p256(nq * n * 4) + p256(L.kjarni_hip_cosine_topk_workspace_bytes(nq, n, k))
off_cand = off; off += p256((4 << 20) * 8)
off_cnt = off; off += p256((nq + 1) * 4)
off_tidx = off; off += p256(nq * k * 8)
off_tsc = off; off += p256(nq * k * 4)
off += p256((nq + 8) * 4)
off_wl = off; off += p256(16 << 20)
off_wc = off
cnt = ws[off_cnt:off_cnt + (nq + 1) * 4].view(torch.int32).cpu()
tsc = ws[off_tsc:off_tsc + nq * 4].view(torch.float32).cpu()
wc = ws[off_wc:off_wc + 3072 * 4].view(torch.int32).cpu()
print("overflow", int(cnt[0]), "final candidates per query: mean", float(cnt[1:].float().mean()), "max", int(cnt[1:].max()))
print("bounds: min %.4f mean %.4f max %.4f; 10th best score mean %.4f" % (float(tsc.min()), float(tsc.mean()), float(tsc.max()), float(sc[:, -1].mean())))
print("wave lists: total", int(wc.sum()), "mean", float(wc.float().mean()), "max", int(wc.max()), "nonzero", int((wc > 0).sum()))
