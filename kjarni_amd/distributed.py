"""Row sharding across the GPUs of one node + all-gather of the output slabs.

The reference is single-process (no communication layer at
all); this is the MI355X-side addition the north star asks for: every sentence /
query-document pair is independent (no cross-row reduction anywhere on the path),
so rows are split into balanced contiguous blocks (floor(N/G) or one more), one process per GPU, weights
replicated, and ONE collective at the end -- an RCCL all-gather of the [N/G, H]
embedding slab (or [N/G] rerank scores) -- leaves the full result on every rank.
xGMI is point-to-point, so the single large all-gather (12.6 MB/rank for the
embed benchmark, 50 KB/rank for rerank) is the only message; nothing is exchanged
inside the layer loop.

torch.distributed is plumbing here (backend "nccl" is RCCL on ROCm; "gloo" on CPU
for the tests); the compute is whatever callable the caller passes (on the GPU
box: HipEncoder.embed_dev / logits_dev)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Balanced contiguous partition: every rank owns floor(n/world) rows and the
    first n % world ranks one more, so no rank idles while another holds two rows'
    worth (n = 9, world = 8 gives 2,1,1,1,1,1,1,1 -- not 2,2,2,2,1,0,0,0)."""
    if world <= 0:
        return 0, n
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gathers row blocks produced under shard_rows() into the full [n_total, ...]
    tensor on every rank.  Blocks differ by at most one row: they are padded to the
    common size for the ONE collective and compacted afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    base, rem = divmod(n_total, world)
    per = base + (1 if rem else 0)
    tail = tuple(local.shape[1:])
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tail, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if rem == 0:
        return out
    # ranks < rem hold `per` rows back to back; the others have one row of padding each
    head = out[:rem * per]
    rest = out[rem * per:].reshape((world - rem, per) + tail)[:, :base].reshape((-1,) + tail)
    return torch.cat([head, rest], dim=0)


def _stream_of(t: torch.Tensor) -> int:
    """Raw HIP stream the encoder should enqueue on: torch's current stream of the tensor's device, so the
    collective that follows (same stream) is ordered behind the kernels; 0 for host tensors (gloo tests)."""
    return int(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else 0


def sharded_map(fn: Callable[[int, int], torch.Tensor], n_total: int, group=None) -> torch.Tensor:
    """Runs fn(start, count) -> [count, ...] on this rank's shard and all-gathers."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = shard_rows(n_total, world, rank)
    return all_gather_rows(fn(start, count), n_total, group)


def rerank_order_arrays(scores: torch.Tensor, top_k: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host-side ordering of gathered rerank scores: stable sort by score descending
    (CrossEncoder::rerank, crates/kjarni-models/src/models/cross_encoder/model.rs:251-252),
    optional truncation (kjarni/src/reranker/model.rs:270-273).  Returns (index int64 [k],
    score f32 [k]) host tensors: `index` = position in the input, as KjarniRerankResult.index."""
    s = scores.detach().to("cpu", torch.float32)
    order = torch.sort(s, descending=True, stable=True).indices
    if top_k is not None:
        order = order[:top_k]
    return order, s[order]


def rerank_order(scores: torch.Tensor, top_k: Optional[int] = None) -> List[Tuple[int, float]]:
    """rerank_order_arrays as a list of (index, score) pairs."""
    idx, sc = rerank_order_arrays(scores, top_k)
    return list(zip(idx.tolist(), sc.tolist()))


def _local_view(t: torch.Tensor, n_total: Optional[int], group) -> Tuple[torch.Tensor, int]:
    """(this rank's rows of t, n_total).  n_total None: t holds the full batch and is sliced; otherwise t IS this
    rank's shard of a batch of n_total rows (weak scaling: no rank ever materialises the other ranks' inputs)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n_total is None:
        start, count = shard_rows(t.shape[0], world, rank)
        return t[start:start + count], t.shape[0]
    _, count = shard_rows(n_total, world, rank)
    if t.shape[0] != count:
        raise ValueError(f"rank {rank} holds {t.shape[0]} rows, its shard of {n_total} rows over {world} ranks is {count}")
    return t, n_total


def sharded_embed(enc, ids: torch.Tensor, mask: torch.Tensor, group=None, n_total: Optional[int] = None
                  ) -> torch.Tensor:
    """ids/mask: int32 [N, S] tensors holding the full batch on every rank, or (with n_total) this rank's shard.
    Every rank encodes its row block (SentenceEncoder::encode_batch_flat semantics: mean pool + L2) and ONE
    all-gather leaves the [N, H] embeddings on every rank."""
    ids_l, n = _local_view(ids, n_total, group)
    mask_l, _ = _local_view(mask, n_total, group)
    count, s = ids_l.shape
    out = torch.empty((count, enc.hidden_size), dtype=torch.float32, device=ids.device)
    if count:
        enc.embed_dev(ids_l.data_ptr(), mask_l.data_ptr(), count, s, out.data_ptr(), stream=_stream_of(ids))
    return all_gather_rows(out, n, group)


def sharded_rerank_scores(enc, ids: torch.Tensor, mask: torch.Tensor, types: torch.Tensor, group=None,
                          n_total: Optional[int] = None) -> torch.Tensor:
    """Pre-tokenised (query, doc) pairs [N, S] -> scores [N] (logit column 0, CrossEncoder::predict_pairs,
    cross_encoder/model.rs:170-240) on every rank; the caller orders them with rerank_order()."""
    ids_l, n = _local_view(ids, n_total, group)
    mask_l, _ = _local_view(mask, n_total, group)
    types_l, _ = _local_view(types, n_total, group)
    count, s = ids_l.shape
    out = torch.empty((count, enc.num_labels), dtype=torch.float32, device=ids.device)
    if count:
        enc.logits_dev(ids_l.data_ptr(), mask_l.data_ptr(), types_l.data_ptr(), count, s, out.data_ptr(),
                       stream=_stream_of(ids))
    return all_gather_rows(out[:, 0].contiguous(), n, group)


def sharded_cosine_topk(local_idx: torch.Tensor, local_score: torch.Tensor, row_offset: int, k: int, group=None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Corpus sharded by rows: every rank holds its local top-k (local row indices,
    scores).  All-gathers the G*k candidates and merges them: score descending,
    ties by ascending GLOBAL index."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    gidx = local_idx.to(torch.int64) + row_offset
    gidx = torch.where(local_idx < 0, torch.full_like(gidx, -1), gidx)
    if world > 1:
        idx_all = torch.empty((world * k,), dtype=torch.int64, device=gidx.device)
        sc_all = torch.empty((world * k,), dtype=torch.float32, device=gidx.device)
        pad_i = torch.full((k,), -1, dtype=torch.int64, device=gidx.device)
        pad_s = torch.full((k,), float("-inf"), dtype=torch.float32, device=gidx.device)
        pad_i[:gidx.numel()] = gidx
        pad_s[:local_score.numel()] = local_score
        dist.all_gather_into_tensor(idx_all, pad_i, group=group)
        dist.all_gather_into_tensor(sc_all, pad_s, group=group)
    else:
        idx_all, sc_all = gidx, local_score
    idx_c, sc_c = idx_all.cpu(), sc_all.cpu()
    keep = idx_c >= 0
    idx_c, sc_c = idx_c[keep], sc_c[keep]
    # lexicographic (score desc, index asc): sort by index first, then stable by score
    o1 = torch.sort(idx_c, stable=True).indices
    idx_c, sc_c = idx_c[o1], sc_c[o1]
    o2 = torch.sort(sc_c, descending=True, stable=True).indices[:k]
    return idx_c[o2], sc_c[o2]


def sharded_cosine_topk_batch(local_idx: torch.Tensor, local_score: torch.Tensor, row_offset: int, k: int, group=None
                              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """The many-query form of sharded_cosine_topk: local_idx / local_score are [nq, k] (what kjarni_hip_cosine_search returns for
    this rank's rows of the corpus; -1 / anything where a query has fewer than k hits).  ONE all-gather of the [nq, k] index
    block and one of the score block (G * nq * k * 12 bytes over xGMI), then every query's G * k candidates are merged on the
    host: score descending, ties by ascending GLOBAL index (the reference's stable sort over the whole corpus,
    vector.rs:150-166).  Returns ([nq, k] int64 global indices, -1 past a query's hits; [nq, k] float32 scores, -inf there)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    assert local_idx.dim() == 2 and local_idx.shape == local_score.shape and local_idx.shape[1] <= k
    nq = local_idx.shape[0]
    gidx = torch.full((nq, k), -1, dtype=torch.int64, device=local_idx.device)
    gsc = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=local_idx.device)
    li = local_idx.to(torch.int64)
    gidx[:, :li.shape[1]] = torch.where(li < 0, torch.full_like(li, -1), li + row_offset)
    gsc[:, :li.shape[1]] = torch.where(li < 0, torch.full_like(local_score, float("-inf")), local_score.to(torch.float32))
    if world > 1:
        idx_all = torch.empty((world * nq, k), dtype=torch.int64, device=gidx.device)     # (rank-major concatenation)
        sc_all = torch.empty((world * nq, k), dtype=torch.float32, device=gidx.device)
        dist.all_gather_into_tensor(idx_all, gidx.contiguous(), group=group)
        dist.all_gather_into_tensor(sc_all, gsc.contiguous(), group=group)
        idx_c = idx_all.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).cpu()
        sc_c = sc_all.view(world, nq, k).permute(1, 0, 2).reshape(nq, world * k).cpu()
    else:
        idx_c, sc_c = gidx.cpu(), gsc.cpu()
    missing = idx_c < 0
    key_i = torch.where(missing, torch.full_like(idx_c, torch.iinfo(torch.int64).max), idx_c)
    o1 = torch.sort(key_i, dim=1, stable=True).indices                   # by global index ...
    idx_c, sc_c, missing = idx_c.gather(1, o1), sc_c.gather(1, o1), missing.gather(1, o1)
    # ... then stable by score: a missing slot sorts after every hit whatever its score field holds (NaN scores keep their place
    # among the hits as the largest key, which is where the scan's own 64-bit keys put them)
    key_s = torch.where(missing, torch.full_like(sc_c, float("-inf")), torch.nan_to_num(sc_c, nan=float("inf")))
    o2 = torch.sort(key_s, dim=1, descending=True, stable=True).indices[:, :k]
    out_i, out_s, out_m = idx_c.gather(1, o2), sc_c.gather(1, o2), missing.gather(1, o2)
    return torch.where(out_m, torch.full_like(out_i, -1), out_i), torch.where(out_m, torch.full_like(out_s, float("-inf")), out_s)
