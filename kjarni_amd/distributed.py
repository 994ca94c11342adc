"""Row sharding across the GPUs of one node + all-gather of the output slabs.

The reference is single-process (no communication layer at
all); this is the MI355X-side addition the north star asks for: every sentence /
query-document pair is independent (no cross-row reduction anywhere on the path),
so rows are split into contiguous blocks of ceil(N/G), one process per GPU, weights
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
    """Contiguous block partition: rank r owns rows [start, start+count) with
    count = ceil(n/world) except for the tail ranks (possibly 0)."""
    per = -(-n // world) if world > 0 else n
    start = min(rank * per, n)
    return start, max(0, min(per, n - start))


def all_gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gathers row blocks produced under shard_rows() into the full [n_total, ...]
    tensor on every rank.  Uneven tails are padded to the common block size for
    the collective and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    per = -(-n_total // world)
    tail = local.shape[1:]
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(tail), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * per,) + tuple(tail), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_total]


def sharded_map(fn: Callable[[int, int], torch.Tensor], n_total: int, group=None) -> torch.Tensor:
    """Runs fn(start, count) -> [count, ...] on this rank's shard and all-gathers."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, count = shard_rows(n_total, world, rank)
    return all_gather_rows(fn(start, count), n_total, group)


def rerank_order(scores: torch.Tensor, top_k: Optional[int] = None) -> List[Tuple[int, float]]:
    """Host-side ordering of gathered rerank scores: stable sort by score
    descending (CrossEncoder::rerank, crates/kjarni-models/src/models/cross_encoder/
    model.rs:251-252), optional truncation (kjarni/src/reranker/model.rs:270-273)."""
    s = scores.detach().to("cpu", torch.float32)
    order = torch.sort(s, descending=True, stable=True).indices.tolist()
    if top_k is not None:
        order = order[:top_k]
    return [(i, float(s[i])) for i in order]


def sharded_embed(enc, ids: torch.Tensor, mask: torch.Tensor, group=None) -> torch.Tensor:
    """ids/mask: int32 [N, S] device tensors holding the FULL batch on every rank
    (or at least this rank's rows).  Returns [N, H] embeddings on every rank."""
    n, s = ids.shape
    h = enc.hidden_size

    def run(start, count):
        out = torch.empty((count, h), dtype=torch.float32, device=ids.device)
        if count:
            stream = torch.cuda.current_stream().cuda_stream
            enc.embed_dev(ids[start:start + count].data_ptr(), mask[start:start + count].data_ptr(), count, s,
                          out.data_ptr(), stream=stream)
        return out

    return sharded_map(run, n, group)


def sharded_rerank_scores(enc, ids: torch.Tensor, mask: torch.Tensor, types: torch.Tensor, group=None
                          ) -> torch.Tensor:
    """Pre-tokenised (query, doc) pairs [N, S] -> scores [N] (logit column 0) on every rank."""
    n, s = ids.shape
    labels = enc.num_labels

    def run(start, count):
        out = torch.empty((count, labels), dtype=torch.float32, device=ids.device)
        if count:
            stream = torch.cuda.current_stream().cuda_stream
            enc.logits_dev(ids[start:start + count].data_ptr(), mask[start:start + count].data_ptr(),
                           types[start:start + count].data_ptr(), count, s, out.data_ptr(), stream=stream)
        return out[:, 0].contiguous()

    return sharded_map(run, n, group)


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
