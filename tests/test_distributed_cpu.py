"""N > 1 path on CPU: world_size-2 gloo processes exercise the row partition,
the (uneven) all-gather, the rerank ordering and the sharded top-k merge with a
deterministic stand-in for the per-row compute, and drive sharded_embed /
sharded_rerank_scores themselves through a host stub with HipEncoder's
interface (the HIP encoder needs a GPU: tests/test_gpu_distributed.py runs the
same functions through it against the oracle)."""
import ctypes as C
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
            counts = [c for _, c in blocks]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    assert D.shard_rows(100000, 8, 7) == (87500, 12500)
    assert [D.shard_rows(9, 8, r)[1] for r in range(8)] == [2, 1, 1, 1, 1, 1, 1, 1]  # no idle rank


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


def _host_u32(ptr, rows, seq):
    return np.ctypeslib.as_array((C.c_uint32 * (rows * seq)).from_address(ptr)).reshape(rows, seq)


class StubEncoder:
    """Stands in for kjarni_amd.HipEncoder on the host (same embed_dev / logits_dev / hidden_size / num_labels):
    raw pointers in, row r of the output depends only on row r of the inputs, so any mis-sharded, mis-ordered or
    dropped row shows up in the gathered result."""
    hidden_size, num_labels = 6, 2

    def embed_dev(self, ids_ptr, mask_ptr, batch, seq, out_ptr, type_ptr=0, pooling=0, normalize=True, fill=0,
                  stream=0):
        assert stream == 0
        ids, mask = _host_u32(ids_ptr, batch, seq), _host_u32(mask_ptr, batch, seq)
        out = np.ctypeslib.as_array((C.c_float * (batch * self.hidden_size)).from_address(out_ptr))
        out.reshape(batch, self.hidden_size)[:] = self.embed_rows(ids, mask)

    def logits_dev(self, ids_ptr, mask_ptr, type_ptr, batch, seq, out_ptr, fill=0, stream=0):
        ids, mask, types = (_host_u32(p, batch, seq) for p in (ids_ptr, mask_ptr, type_ptr))
        out = np.ctypeslib.as_array((C.c_float * (batch * self.num_labels)).from_address(out_ptr))
        out.reshape(batch, self.num_labels)[:] = self.logit_rows(ids, mask, types)

    @classmethod
    def embed_rows(cls, ids, mask):
        x = (ids.astype(np.float64) * mask).sum(1)
        return np.stack([np.sin(x * (j + 1) * 1e-3) for j in range(cls.hidden_size)], 1).astype(np.float32)

    @classmethod
    def logit_rows(cls, ids, mask, types):
        x = (ids.astype(np.float64) * mask * (1 + types)).sum(1)
        return np.stack([np.cos(x * 1e-3), np.sin(x * 1e-3)], 1).astype(np.float32)


def _token_batch(n, seq=8):
    rng = np.random.default_rng(n)
    ids = rng.integers(1, 30000, (n, seq), dtype=np.int64).astype(np.uint32)
    mask = (rng.random((n, seq)) < 0.8).astype(np.uint32)
    types = (rng.random((n, seq)) < 0.5).astype(np.uint32)
    t = lambda a: torch.from_numpy(a.view(np.int32))  # noqa: E731
    return (ids, mask, types), (t(ids), t(mask), t(types))


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
            # the product entry points themselves, full batch on every rank ...
            enc = StubEncoder()
            (ids, mask, types), (tids, tmask, ttypes) = _token_batch(n)
            want_e = torch.from_numpy(enc.embed_rows(ids, mask)) if n else torch.zeros((0, enc.hidden_size))
            want_s = torch.from_numpy(enc.logit_rows(ids, mask, types)[:, 0]) if n else torch.zeros((0,))
            got_e = D.sharded_embed(enc, tids, tmask)
            got_s = D.sharded_rerank_scores(enc, tids, tmask, ttypes)
            ok &= got_e.shape == want_e.shape and bool(torch.equal(got_e, want_e))
            ok &= got_s.shape == want_s.shape and bool(torch.equal(got_s, want_s))
            # ... and with every rank holding only its own shard (the weak-scaling bench's form)
            st, cnt = D.shard_rows(n, world, rank)
            sl = slice(st, st + cnt)
            got_e = D.sharded_embed(enc, tids[sl].contiguous(), tmask[sl].contiguous(), n_total=n)
            got_s = D.sharded_rerank_scores(enc, tids[sl].contiguous(), tmask[sl].contiguous(),
                                            ttypes[sl].contiguous(), n_total=n)
            ok &= bool(torch.equal(got_e, want_e)) and bool(torch.equal(got_s, want_s))
            ok &= D.rerank_order(got_s) == D.rerank_order(want_s)
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
        # the many-query form: [nq, k] local lists (one query with ties across the shard boundary, one with fewer than k hits on a
        # rank), ONE all-gather per array, merged per query by (score descending, global index ascending)
        nq = 5
        g = torch.Generator().manual_seed(7)
        full = torch.randn((nq, 1001), generator=g)
        full[1] = torch.round(full[1] * 2) / 2                       # many exact ties, also across ranks
        loc2 = full[:, start:start + count]
        o2 = torch.sort(loc2, dim=1, descending=True, stable=True).indices[:, :k]
        li, ls = o2.clone(), loc2.gather(1, o2)
        if rank == world - 1:
            li[3, 2:], ls[3, 2:] = -1, 123.0                         # this rank found only two hits for query 3
        bi, bs = D.sharded_cosine_topk_batch(li, ls, start, k)
        for j in range(nq):
            row = full[j].clone()
            if j == 3:                                               # the last rank's rows beyond its two best do not exist
                s2, c2 = D.shard_rows(1001, world, world - 1)
                keep = torch.sort(row[s2:s2 + c2], descending=True, stable=True).indices[:2] + s2
                m = torch.ones(1001, dtype=torch.bool)
                m[s2:s2 + c2] = False
                m[keep] = True
                row[~m] = float("-inf")
            ref = torch.sort(row, descending=True, stable=True).indices[:k]
            ok &= bi[j].tolist() == ref.tolist() and bool(torch.equal(bs[j], row[ref]))
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


def test_world_size_3_uneven_rows_gloo():
    """An N that no world size divides (100 003 rows over 3 ranks: 33 335 + 33 334 + 33 334): blocks padded to the common size for the
    ONE collective and compacted after it, through the product entry points (full batch on every rank and shard only)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, [2, 100003], q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True), (2, True)]


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("rows", [7, 16, 100003, 0])
def test_in_process_group_gather_plan(n, rows):
    """EncoderGroup::gather (csrc/group.cpp) issues its RCCL calls from EncoderGroup::gather_plan, pure host arithmetic exported
    as kjarni_hip_group_gather_plan: equal blocks -> ONE in-place ncclAllGather per rank; blocks that differ by a row (N % n != 0:
    the path a `--in-process` run with 100 003 rows takes) -> one ncclBroadcast per non-empty block and rank.  The plan is executed
    here on host arrays with the collectives' semantics: every rank must end with every block at its place, every call of one
    collective must agree on root / count across ranks, and no call may touch floats outside its block."""
    import ctypes as C

    import numpy as np

    from kjarni_amd import _ffi

    class Op(C.Structure):
        _fields_ = [("rank", C.c_int32), ("root", C.c_int32), ("offset", C.c_int64), ("floats", C.c_int64)]

    L = _ffi.lib()
    width = 5
    ops = (Op * (n * n + 1))()
    cnt = C.c_size_t(0)
    assert L.kjarni_hip_group_gather_plan(rows, n, width, C.cast(ops, C.c_void_p), len(ops), C.byref(cnt)) == 0
    plan = [(o.rank, o.root, o.offset, o.floats) for o in ops[:cnt.value]]
    shards = [D.shard_rows(rows, n, i) for i in range(n)]          # the same partition as EncoderGroup::shard
    assert sum(c for _, c in shards) == rows and all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(n - 1))
    # every rank's buffer holds ITS block (value = 1000 * rank + position) and NaN elsewhere
    bufs = []
    for i, (s, c) in enumerate(shards):
        b = np.full(rows * width, np.nan, np.float64)
        b[s * width:(s + c) * width] = 1000.0 * i + np.arange(c * width)
        bufs.append(b)
    even = rows % n == 0
    per_rank = [[p for p in plan if p[0] == i] for i in range(n)]
    if even:
        assert all(len(p) == 1 and p[0][1] == -1 for p in per_rank)
        counts = {p[0][3] for p in per_rank}
        assert len(counts) == 1                                      # ncclAllGather: the same sendcount on every rank
        for i, (s, c) in enumerate(shards):
            assert per_rank[i][0][2] == s * width and per_rank[i][0][3] == c * width    # in place: send = own slot of recv
        sends = [bufs[i][per_rank[i][0][2]:per_rank[i][0][2] + per_rank[i][0][3]].copy() for i in range(n)]
        for b in bufs:
            b[:] = np.concatenate(sends) if rows else b
    else:
        seqs = [[(p[1], p[3]) for p in pr] for pr in per_rank]
        assert all(sq == seqs[0] for sq in seqs)                     # every rank issues the same broadcasts in the same order
        assert [r for r, _ in seqs[0]] == [i for i, (_, c) in enumerate(shards) if c > 0]
        for step in range(len(seqs[0])):
            root = per_rank[0][step][1]
            s, c = shards[root]
            assert all(pr[step][2] == s * width and pr[step][3] == c * width for pr in per_rank)
            data = bufs[root][s * width:(s + c) * width].copy()
            assert not np.isnan(data).any()
            for b in bufs:
                b[s * width:(s + c) * width] = data
    want = np.concatenate([1000.0 * i + np.arange(c * width) for i, (s, c) in enumerate(shards)]) if rows else np.zeros(0)
    for b in bufs:
        assert np.array_equal(b, want)
    # bad arguments are refused, a short output array still reports the full count
    assert L.kjarni_hip_group_gather_plan(-1, n, width, None, 0, C.byref(cnt)) == 7
    assert L.kjarni_hip_group_gather_plan(rows, 0, width, None, 0, C.byref(cnt)) == 7
    assert L.kjarni_hip_group_gather_plan(rows, n, width, None, 0, C.byref(cnt)) == 0 and cnt.value == len(plan)
