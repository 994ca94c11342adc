"""N > 1 path on CPU: world_size-2 gloo processes exercise the row partition,
the (uneven) all-gather, the rerank ordering and the sharded top-k merge with a
deterministic stand-in for the per-row compute (the HIP encoder itself needs a
GPU; its parity is covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kjarni_amd import distributed as D


def test_shard_rows_partition():
    for n in (0, 1, 7, 8, 9, 100000, 65536):
        for world in (1, 2, 3, 8):
            blocks = [D.shard_rows(n, world, r) for r in range(world)]
            assert sum(c for _, c in blocks) == n
            pos = 0
            for s, c in blocks:
                assert s == min(pos, n)
                pos += c
            per = -(-n // world)
            assert all(c <= per for _, c in blocks)
    assert D.shard_rows(100000, 8, 7) == (87500, 12500)


def test_rerank_order_is_stable_descending():
    s = torch.tensor([0.5, 2.0, 0.5, -1.0, 2.0])
    assert [i for i, _ in D.rerank_order(s)] == [1, 4, 0, 2, 3]
    assert [i for i, _ in D.rerank_order(s, top_k=2)] == [1, 4]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_rows(start, count, width):
    """Row r of the 'model output' depends only on r."""
    r = torch.arange(start, start + count, dtype=torch.float32)
    return torch.stack([torch.sin(r * (j + 1)) for j in range(width)], dim=1) if width else r


def _worker(rank, world, port, n_list, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for n in n_list:
            full = _fake_rows(0, n, 5)
            got = D.sharded_map(lambda s, c: _fake_rows(s, c, 5), n)
            ok &= got.shape == full.shape and bool(torch.equal(got, full))
            # rerank: scores gathered, every rank derives the same order
            sc = D.sharded_map(lambda s, c: _fake_rows(s, c, 0).mul(0.37).cos(), n)
            ref = (_fake_rows(0, n, 0) * 0.37).cos()
            ok &= bool(torch.equal(sc, ref))
            ok &= D.rerank_order(sc, 5) == D.rerank_order(ref, 5)
        # sharded cosine top-k merge: corpus rows split, local top-k, global merge
        rng = np.random.default_rng(0)
        scores = torch.from_numpy(rng.standard_normal(1001).astype(np.float32))
        scores[10] = scores[900] = 5.0   # a tie across shards: lower global index first
        start, count = D.shard_rows(1001, world, rank)
        k = 7
        loc = scores[start:start + count]
        o = torch.sort(loc, descending=True, stable=True).indices[:k]
        idx, sc = D.sharded_cosine_topk(o, loc[o], start, k)
        ref_o = torch.sort(scores, descending=True, stable=True).indices[:k]
        ok &= idx.tolist() == ref_o.tolist() and bool(torch.equal(sc, scores[ref_o]))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, [0, 1, 9, 10, 257], q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
