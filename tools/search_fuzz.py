"""Differential fuzz of kjarni_hip_cosine_search's many-query routes against kjarni_hip_cosine_scores + kjarni_hip_cosine_topk
(the matrix-core scan: queries padded to >= 20 rows so that the two-call form takes it): random corpus sizes, widths, query
counts, k, both zero-norm conventions, duplicated rows (exact ties), zero rows, a zero query.  Indices and score bits must agree.
usage: python tools/search_fuzz.py [cases, default 40] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kjarni_amd import _ffi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = _ffi.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
bad = 0
for case in range(cases):
    dim = int(rng.choice([128, 256, 384, 512, 640, 768, 1024]))
    n = int(rng.integers(20_000, 400_000 if dim <= 512 else 150_000))
    nq = int(rng.choice([2, 3, 5, 17, 19, 20, 33, 64, 65, 130]))
    k = int(rng.choice([1, 3, 10, 12, 100, 129, 300]))
    mode = int(rng.integers(0, 2))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    corpus = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
    q = torch.randn((nq, dim), generator=g, device=dev, dtype=torch.float32)
    flavour = int(rng.integers(0, 4))
    if flavour == 1:                       # exact ties: the second half repeats the first
        corpus[n // 2:] = corpus[: n - n // 2].clone()
    if flavour == 2:                       # scores ascend with the index for one query (the sample under-estimates its bound)
        corpus += torch.linspace(0.0, 1.5, n, device=dev)[:, None] * q[nq // 2][None, :]
    if flavour == 3 and nq > 2:            # a zero query, zero rows, tiny rows
        q[1] = 0.0
        corpus[n // 3] = 0.0
        corpus[n // 5] *= 1e-7
    # (widths without the filter pass: below 20 queries the search IS the two calls, with the streaming passes' arithmetic)
    pad = nq if dim == 640 and nq < 20 else max(nq, 20)
    qp = torch.zeros((pad, dim), device=dev, dtype=torch.float32)
    qp[:nq] = q
    qp[nq:] = q[0]
    scores = torch.empty((pad, n), dtype=torch.float32, device=dev)
    ws = torch.empty(L.kjarni_hip_cosine_topk_workspace_bytes(pad, n, k), dtype=torch.uint8, device=dev)
    idx2 = torch.empty((pad, k), dtype=torch.int64, device=dev)
    sc2 = torch.empty((pad, k), dtype=torch.float32, device=dev)
    _ffi.check_error(L.kjarni_hip_cosine_scores(0, qp.data_ptr(), pad, corpus.data_ptr(), n, dim, mode, scores.data_ptr(), st))
    _ffi.check_error(L.kjarni_hip_cosine_topk(0, scores.data_ptr(), pad, n, k, ws.data_ptr(), idx2.data_ptr(), sc2.data_ptr(), st))
    del scores, ws
    ws1 = torch.empty(L.kjarni_hip_cosine_search_workspace_bytes(nq, n, dim, k), dtype=torch.uint8, device=dev)
    idx1 = torch.full((nq, k), -7, dtype=torch.int64, device=dev)
    sc1 = torch.full((nq, k), 7.0, dtype=torch.float32, device=dev)
    _ffi.check_error(L.kjarni_hip_cosine_search(0, q.data_ptr(), nq, corpus.data_ptr(), n, dim, mode, k, ws1.data_ptr(), idx1.data_ptr(),
                                                sc1.data_ptr(), st))
    torch.cuda.synchronize()
    same_i = bool(torch.equal(idx1, idx2[:nq]))
    same_s = bool(torch.equal(sc1.view(torch.int32), sc2[:nq].view(torch.int32)))
    print(f"case {case:3d}: n {n:7d} dim {dim:4d} nq {nq:3d} k {k:3d} mode {mode} flavour {flavour}: indices {'ok' if same_i else 'DIFFER'}, "
          f"scores {'ok' if same_s else 'DIFFER'}", flush=True)
    if not (same_i and same_s):
        a, b = sc1.cpu().numpy(), sc2[:nq].cpu().numpy()
        ia, ib = idx1.cpu().numpy(), idx2[:nq].cpu().numpy()
        rows = sorted(set(np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1) | (ia != ib).any(axis=1))[0].tolist()))
        print(f"          queries {rows[:10]}; max |score difference| {np.nanmax(np.abs(a - b)):.3e}; first differing row: search {a[rows[0]][:6]} "
              f"{ia[rows[0]][:6]} two-call {b[rows[0]][:6]} {ib[rows[0]][:6]}", flush=True)
    bad += 0 if (same_i and same_s) else 1
    del ws1, corpus, q, qp
    torch.cuda.empty_cache()
print(f"{cases - bad} / {cases} cases agree bit for bit")
sys.exit(1 if bad else 0)
